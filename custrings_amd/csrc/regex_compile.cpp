// Host-side regex compiler: UTF-8 pattern -> csrx::Program.
//
// Behavioural contract = the reference's Reprog builder
// (/root/reference/cpp/src/regex/regcomp.cpp): lexer :314-539, bracket
// classes :170-312, counted-repeat expansion :772-885, operator-precedence
// assembly :609-767, NOP elision :968-1037, leading-alternation start list
// :1040-1061.  The executor's match priorities are defined on instruction
// order, so the emitted stream has to be identical, including what happens on
// malformed input.  tests/test_regex_compile.py diffs this compiler against
// the reference compiler built in place into oracle/_ref.
#include <cstdio>
#include <cstring>

#include "regex_program.h"

namespace csrx {
namespace {

// Operator tokens carry their precedence in their value; bits 0300 == 0200
// marks "operator".  Operands are everything else.
enum Tok : int {
  T_START = 0200,
  T_RBRA = OP_RBRA,
  T_LBRA = OP_LBRA,
  T_LBRA_NC = 0203,
  T_OR = OP_OR,
  T_CAT = 0205,
  T_STAR = 0206,
  T_STAR_LAZY = 0207,
  T_PLUS = 0210,
  T_PLUS_LAZY = 0211,
  T_QUEST = 0212,
  T_QUEST_LAZY = 0213,
  T_COUNTED = 0214,
  T_COUNTED_LAZY = 0215,
  T_NOP = 0302,
};
inline bool is_operator(int t) { return (t & 0300) == 0200; }

struct Token {
  int kind = 0;
  uint32_t ch = 0;   // CHAR / BOL / EOL payload
  int cls = 0;       // class id for CCLASS / NCCLASS
  short lo = 0, hi = 0;  // counted repeat bounds, hi < 0 = unbounded
};

// ---------------------------------------------------------------- lexer ----
class Lexer {
 public:
  Lexer(const uint32_t* pat, Program& prog) : p_(pat), prog_(prog) {
    nul_ = pat;
    while (*nul_) ++nul_;
  }

  bool has_counted = false;

  std::vector<Token> run() {
    std::vector<Token> out;
    for (;;) {
      Token t = next();
      if (t.kind == OP_END) break;
      if (t.kind == T_COUNTED || t.kind == T_COUNTED_LAZY) has_counted = true;
      out.push_back(t);
    }
    return out;
  }

 private:
  const uint32_t* p_;
  const uint32_t* nul_;
  Program& prog_;
  bool done_ = false;
  int id_w_ = -1, id_W_ = -1, id_s_ = -1, id_d_ = -1, id_D_ = -1;

  // reads never run past the terminator: beyond it the pattern reads as NULs
  uint32_t get() { return p_ > nul_ ? 0u : *p_++; }
  uint32_t peek(int k = 0) const { return p_ + k > nul_ ? 0u : p_[k]; }

  // returns true when the character was backslash-escaped
  bool read(uint32_t& c) {
    if (done_) {
      c = 0;
      return true;
    }
    c = get();
    if (c == '\\') {
      c = get();
      return true;
    }
    if (c == 0) done_ = true;
    return false;
  }

  int shared_class(int& slot, int builtins, bool with_newline) {
    if (slot < 0) {
      CharClass cc;
      cc.builtins = builtins;
      if (with_newline) {
        cc.ranges.push_back('\n');
        cc.ranges.push_back('\n');
      }
      prog_.classes.push_back(cc);
      slot = (int)prog_.classes.size() - 1;
    }
    return slot;
  }

  static int hex_hi(uint32_t a) {
    // note the strict '>' on letters: 'a'/'A' themselves contribute nothing
    if (a >= '0' && a <= '9') return (int)(a - '0');
    if (a > 'a' && a <= 'f') return (int)(a - 'a' + 10);
    if (a > 'A' && a <= 'F') return (int)(a - 'A' + 10);
    return 0;
  }

  static void scan_short(const char* buf, short& v) { sscanf(buf, "%hd", &v); }

  Token bracket_class() {
    Token t;
    t.kind = OP_CCLASS;
    std::vector<uint32_t> spans;
    int builtins = 0;
    uint32_t c;
    bool esc = read(c);
    if (!esc && c == '^') {
      t.kind = OP_NCCLASS;
      esc = read(c);
      // negated classes never match newline
      spans.push_back('\n');
      spans.push_back('\n');
    }
    for (int seen = 1;; ++seen) {
      if (c == 0) {  // unterminated class
        t.kind = 0;
        return t;
      }
      if (esc) {
        int bi = 0;
        switch (c) {
          case 'n': c = '\n'; break;
          case 'r': c = '\r'; break;
          case 't': c = '\t'; break;
          case 'a': c = 0x07; break;
          case 'b': c = 0x08; break;
          case 'f': c = 0x0C; break;
          case 'w': bi = BI_w; break;
          case 's': bi = BI_s; break;
          case 'd': bi = BI_d; break;
          case 'W': bi = BI_W; break;
          case 'S': bi = BI_S; break;
          case 'D': bi = BI_D; break;
        }
        if (bi) {
          builtins |= bi;
          esc = read(c);
          continue;
        }
      }
      if (!esc && c == ']' && seen > 1) break;
      if (!esc && c == '-') {
        if (spans.empty()) {
          t.kind = 0;
          return t;
        }
        esc = read(c);
        if ((!esc && c == ']') || c == 0) {
          t.kind = 0;
          return t;
        }
        spans.back() = c;
      } else {
        spans.push_back(c);
        spans.push_back(c);
      }
      esc = read(c);
    }
    // order spans by start with the same exchange sort the contract uses, so
    // ties between equal starts resolve identically
    for (size_t a = 0; a < spans.size(); a += 2)
      for (size_t b = a + 2; b < spans.size(); b += 2)
        if (spans[b] < spans[a]) {
          std::swap(spans[a], spans[b]);
          std::swap(spans[a + 1], spans[b + 1]);
        }
    CharClass cc;
    cc.builtins = builtins;
    if (spans.size() >= 2) {
      cc.ranges.push_back(spans[0]);
      cc.ranges.push_back(spans[1]);
      for (size_t a = 2; a < spans.size(); a += 2) {
        uint32_t& cur_hi = cc.ranges.back();
        if (spans[a] <= cur_hi + 1) {  // overlapping or adjacent
          if (spans[a + 1] >= cur_hi) cur_hi = spans[a + 1];
        } else {
          cc.ranges.push_back(spans[a]);
          cc.ranges.push_back(spans[a + 1]);
        }
      }
    }
    prog_.classes.push_back(cc);
    t.cls = (int)prog_.classes.size() - 1;
    return t;
  }

  bool counted(Token& t) {
    if (peek() < '0' || peek() > '9') return false;
    const uint32_t* rewind = p_;
    char buf[8];
    buf[0] = 0;
    for (int i = 0; i < 7 && peek() != '}' && peek() != ',' && peek() != 0; ++i) {
      buf[i] = (char)get();
      buf[i + 1] = 0;
    }
    if (peek() != '}' && peek() != ',') {
      p_ = rewind;
      return false;
    }
    scan_short(buf, t.lo);
    if (peek() != ',') {
      t.hi = t.lo;
    } else {
      t.hi = -1;
      get();
      buf[0] = 0;
      for (int i = 0; i < 7 && peek() != '}' && peek() != 0; ++i) {
        buf[i] = (char)get();
        buf[i + 1] = 0;
      }
      if (peek() != '}') {
        p_ = rewind;
        return false;
      }
      if (buf[0] != 0) scan_short(buf, t.hi);
    }
    get();  // '}'
    t.kind = T_COUNTED;
    if (peek() == '?') {
      get();
      t.kind = T_COUNTED_LAZY;
    }
    return true;
  }

  int lazy_if_quest(int greedy, int lazy) {
    if (peek() == '?') {
      get();
      return lazy;
    }
    return greedy;
  }

  Token next() {
    Token t;
    uint32_t c;
    bool esc = read(c);
    t.ch = c;
    t.kind = OP_CHAR;
    if (esc) {
      if (c == 0) {
        t.kind = OP_END;
        return t;
      }
      if (c >= '0' && c <= '7') {
        // every escaped digit string is octal; the character after the digits
        // is consumed as well (contract quirk, regcomp.cpp:322-333)
        uint32_t v = c - '0';
        uint32_t d = get();
        while (d >= '0' && d <= '7') {
          v = (v << 3) | (d - '0');
          d = get();
        }
        t.ch = v;
        return t;
      }
      switch (c) {
        case 't': t.ch = '\t'; break;
        case 'n': t.ch = '\n'; break;
        case 'r': t.ch = '\r'; break;
        case 'a': t.ch = 0x07; break;
        case 'f': t.ch = 0x0C; break;
        case 'x': {
          uint32_t a = get();
          uint32_t b = get();
          t.ch = (uint32_t)((hex_hi(a) << 4) + hex_hi(b));
          break;
        }
        case 'w': t.kind = OP_CCLASS; t.cls = shared_class(id_w_, BI_w, false); break;
        case 'W': t.kind = OP_NCCLASS; t.cls = shared_class(id_W_, BI_w, true); break;
        case 's': t.kind = OP_CCLASS; t.cls = shared_class(id_s_, BI_s, false); break;
        case 'S': t.kind = OP_NCCLASS; t.cls = shared_class(id_s_, BI_s, false); break;
        case 'd': t.kind = OP_CCLASS; t.cls = shared_class(id_d_, BI_d, false); break;
        case 'D': t.kind = OP_NCCLASS; t.cls = shared_class(id_D_, BI_d, true); break;
        case 'b': t.kind = OP_BOW; break;
        case 'B': t.kind = OP_NBOW; break;
        case 'A': t.kind = OP_BOL; break;
        case 'Z': t.kind = OP_EOL; break;
      }
      return t;
    }
    switch (c) {
      case 0: t.kind = OP_END; break;
      case '*': t.kind = lazy_if_quest(T_STAR, T_STAR_LAZY); break;
      case '?': t.kind = lazy_if_quest(T_QUEST, T_QUEST_LAZY); break;
      case '+': t.kind = lazy_if_quest(T_PLUS, T_PLUS_LAZY); break;
      case '{': counted(t); break;
      case '|': t.kind = T_OR; break;
      case '.': t.kind = OP_ANY; break;
      case '(':
        t.kind = T_LBRA;
        if (peek() == '?' && peek(1) == ':') {
          get();
          get();
          t.kind = T_LBRA_NC;
        }
        break;
      case ')': t.kind = T_RBRA; break;
      case '^': t.kind = OP_BOL; break;
      case '$': t.kind = OP_EOL; break;
      case '[': {
        Token b = bracket_class();
        b.ch = c;
        return b;
      }
    }
    return t;
  }
};

// ------------------------------------------- counted-repeat expansion ----
// a{n,m} is rewritten into plain tokens before assembly: n copies, then
// (m-n) nested optional non-capturing groups, or +/* when unbounded.
std::vector<Token> expand_counted(const std::vector<Token>& in) {
  std::vector<Token> out;
  std::vector<int> open;
  int unit = -1;  // index in `in` where the repeated unit starts
  auto simple = [](int kind) {
    Token t;
    t.kind = kind;
    return t;
  };
  for (int i = 0; i < (int)in.size(); ++i) {
    const Token& tk = in[i];
    if (tk.kind != T_COUNTED && tk.kind != T_COUNTED_LAZY) {
      out.push_back(tk);
      if (tk.kind == T_LBRA || tk.kind == T_LBRA_NC) {
        open.push_back(i);
        unit = -1;
      } else if (tk.kind == T_RBRA) {
        // an unbalanced ')' has no opener to point at; treat as broken
        if (open.empty()) {
          unit = -1;
        } else {
          unit = open.back();
          open.pop_back();
        }
      } else if (!is_operator(tk.kind)) {
        unit = i;
      }
      continue;
    }
    if (unit < 0) return out;  // nothing to repeat: stop here
    const bool lazy = tk.kind == T_COUNTED_LAZY;
    if (tk.lo <= 0) {
      for (int j = 0; j < i - unit && !out.empty(); ++j) out.pop_back();
    } else {
      for (int rep = 1; rep < tk.lo; ++rep)
        for (int k = unit; k < i; ++k) out.push_back(in[k]);
    }
    if (tk.hi >= 0) {
      for (int rep = tk.lo; rep < tk.hi; ++rep) {
        out.push_back(simple(T_LBRA_NC));
        for (int k = unit; k < i; ++k) out.push_back(in[k]);
      }
      for (int rep = tk.lo; rep < tk.hi; ++rep) {
        out.push_back(simple(T_RBRA));
        out.push_back(simple(lazy ? T_QUEST_LAZY : T_QUEST));
      }
    } else if (tk.lo > 0) {
      out.push_back(simple(lazy ? T_PLUS_LAZY : T_PLUS));
    } else {
      for (int k = unit; k < i; ++k) out.push_back(in[k]);
      out.push_back(simple(lazy ? T_STAR_LAZY : T_STAR));
    }
  }
  return out;
}

// ------------------------------------------------------------ assembler ----
// Operator-precedence assembly of fragments {first,last}; instruction ids are
// allocated in reduction order, which fixes the numbering.
class Assembler {
 public:
  explicit Assembler(Program& prog) : prog_(prog) {}

  void build(const std::vector<Token>& toks) {
    push_op(T_START - 1, 0);
    for (const Token& tk : toks) {
      cur_ = tk;
      int kind = tk.kind;
      if (kind == T_LBRA) {
        push_group_ = ++groups_;
      } else if (kind == T_LBRA_NC) {
        push_group_ = 0;
        kind = T_LBRA;
      }
      if (is_operator(kind))
        on_operator(kind);
      else
        on_operand(kind);
    }
    reduce_until(T_START);
    cur_ = Token();
    on_operand(OP_END);
    reduce_until(T_START);
    prog_.start_inst = frags_.back().first;
    prog_.num_groups = groups_;
  }

 private:
  struct Frag { int first, last; };
  struct PendingOp { int kind, group; };

  Program& prog_;
  std::vector<Frag> frags_;
  std::vector<PendingOp> ops_;
  Token cur_;
  bool prev_operand_ = false;
  int depth_ = 0, groups_ = 0, push_group_ = 0;

  int emit(int type) {
    prog_.insts.push_back(Inst{type, 0, 0});
    return (int)prog_.insts.size() - 1;
  }
  Inst& at(int id) { return prog_.insts[id]; }
  void push_op(int kind, int group) { ops_.push_back({kind, group}); }

  Frag pop_frag() {
    if (frags_.empty()) {  // operator without operand: invent a no-op
      int id = emit(T_NOP);
      frags_.push_back({id, id});
    }
    Frag f = frags_.back();
    frags_.pop_back();
    return f;
  }

  void reduce_until(int pri) {
    while (pri == T_RBRA || ops_.back().kind >= pri) {
      PendingOp op = ops_.back();
      ops_.pop_back();
      switch (op.kind) {
        default: break;
        case T_LBRA: {  // closes a group
          Frag body = pop_frag();
          int close = emit(OP_RBRA);
          at(close).u1 = op.group;
          at(body.last).u2 = close;
          int open = emit(OP_LBRA);
          at(open).u1 = op.group;
          at(open).u2 = body.first;
          frags_.push_back({open, close});
          return;
        }
        case T_OR: {
          Frag rhs = pop_frag();
          Frag lhs = pop_frag();
          int join = emit(T_NOP);
          at(rhs.last).u2 = join;
          at(lhs.last).u2 = join;
          int fork = emit(OP_OR);
          at(fork).u1 = lhs.first;  // preferred branch
          at(fork).u2 = rhs.first;
          frags_.push_back({fork, join});
          break;
        }
        case T_CAT: {
          Frag rhs = pop_frag();
          Frag lhs = pop_frag();
          at(lhs.last).u2 = rhs.first;
          frags_.push_back({lhs.first, rhs.last});
          break;
        }
        case T_STAR: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          at(body.last).u2 = fork;
          at(fork).u1 = body.first;
          frags_.push_back({fork, fork});
          break;
        }
        case T_STAR_LAZY: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          int exit = emit(T_NOP);
          at(body.last).u2 = fork;
          at(fork).u2 = body.first;
          at(fork).u1 = exit;
          frags_.push_back({fork, exit});
          break;
        }
        case T_PLUS: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          at(body.last).u2 = fork;
          at(fork).u1 = body.first;
          frags_.push_back({body.first, fork});
          break;
        }
        case T_PLUS_LAZY: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          int exit = emit(T_NOP);
          at(body.last).u2 = fork;
          at(fork).u2 = body.first;
          at(fork).u1 = exit;
          frags_.push_back({body.first, exit});
          break;
        }
        case T_QUEST: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          int exit = emit(T_NOP);
          at(fork).u2 = exit;
          at(fork).u1 = body.first;
          at(body.last).u2 = exit;
          frags_.push_back({fork, exit});
          break;
        }
        case T_QUEST_LAZY: {
          Frag body = pop_frag();
          int fork = emit(OP_OR);
          int exit = emit(T_NOP);
          at(fork).u2 = body.first;
          at(fork).u1 = exit;
          at(body.last).u2 = exit;
          frags_.push_back({fork, exit});
          break;
        }
      }
    }
  }

  void on_operator(int kind) {
    if (kind == T_RBRA && --depth_ < 0) return;  // unmatched ')'
    if (kind == T_LBRA) {
      ++depth_;
      if (prev_operand_) on_operator(T_CAT);
    } else {
      reduce_until(kind);
    }
    if (kind != T_RBRA) push_op(kind, push_group_);
    prev_operand_ = kind == T_STAR || kind == T_QUEST || kind == T_PLUS ||
                    kind == T_STAR_LAZY || kind == T_QUEST_LAZY ||
                    kind == T_PLUS_LAZY || kind == T_RBRA;
  }

  void on_operand(int kind) {
    if (prev_operand_) on_operator(T_CAT);  // implicit concatenation
    int id = emit(kind);
    if (kind == OP_CCLASS || kind == OP_NCCLASS)
      at(id).u1 = cur_.cls;
    else if (kind == OP_CHAR || kind == OP_BOL || kind == OP_EOL)
      at(id).u1 = (int32_t)cur_.ch;
    frags_.push_back({id, id});
    prev_operand_ = true;
  }
};

// ------------------------------------------------------------- cleanup ----
void strip_nops(Program& prog) {
  auto& in = prog.insts;
  const int n = (int)in.size();
  // non-capturing brackets are no-ops
  for (auto& i : in)
    if ((i.type == OP_LBRA || i.type == OP_RBRA) && i.u1 < 1) i.type = T_NOP;
  // (a dangling repeat such as "*a" makes a NOP chain loop back on itself in
  // the reference, which then never terminates; bound the walk instead)
  auto skip = [&](int id) {
    for (int hops = 0; in[id].type == T_NOP && hops <= n; ++hops) id = in[id].u2;
    return id;
  };
  for (int i = 0; i < n; ++i) {
    if (in[i].type == T_NOP) continue;
    in[i].u2 = skip(in[i].u2);
    if (in[i].type == OP_OR) in[i].u1 = skip(in[i].u1);
  }
  prog.start_inst = skip(prog.start_inst);
  std::vector<int> remap(n);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    remap[i] = kept;
    if (in[i].type != T_NOP) in[kept++] = in[i];
  }
  in.resize(kept);
  for (auto& i : in) {
    i.u2 = remap[i.u2];
    if (i.type == OP_OR) i.u1 = remap[i.u1];
  }
  prog.start_inst = remap[prog.start_inst];
}

// A leading alternation is unrolled into several seed instructions.
void collect_starts(Program& prog) {
  prog.starts.clear();
  std::vector<int> todo{prog.start_inst};
  std::vector<char> seen(prog.insts.size(), 0);  // self-referential ORs (see strip_nops)
  while (!todo.empty()) {
    int id = todo.back();
    todo.pop_back();
    const Inst& i = prog.insts[id];
    if (i.type == OP_OR) {
      if (seen[id]) continue;
      seen[id] = 1;
      todo.push_back(i.u2);
      todo.push_back(i.u1);
    } else {
      prog.starts.push_back(id);
    }
  }
  prog.starts.push_back(-1);
}

}  // namespace

std::vector<uint32_t> pack_utf8(const char* s) {
  std::vector<uint32_t> out;
  const unsigned char* p = (const unsigned char*)s;
  while (*p) {
    unsigned b = *p;
    int w = 1 + ((b & 0xF0) == 0xF0) + ((b & 0xE0) == 0xE0) + ((b & 0xC0) == 0xC0) -
            ((b & 0xC0) == 0x80);
    uint32_t c = b;
    ++p;
    for (int k = 1; k < w && *p; ++k) c = (c << 8) | *p++;
    out.push_back(c);
  }
  out.push_back(0);
  return out;
}

Program compile(const uint32_t* pattern) {
  Program prog;
  Lexer lx(pattern, prog);
  std::vector<Token> toks = lx.run();
  if (lx.has_counted) toks = expand_counted(toks);
  Assembler(prog).build(toks);
  strip_nops(prog);
  collect_starts(prog);
  return prog;
}

std::vector<int32_t> Program::to_blob() const {
  std::vector<int32_t> b(kBlobHeaderWords, 0);
  b[0] = kBlobMagic;
  b[1] = start_inst;
  b[2] = num_groups;
  b[3] = (int32_t)insts.size();
  b[4] = (int32_t)starts.size();
  b[5] = (int32_t)classes.size();
  for (const Inst& i : insts) {
    b.push_back(i.type);
    b.push_back(i.u1);
    b.push_back(i.u2);
    b.push_back(0);
  }
  for (int32_t s : starts) b.push_back(s);
  int32_t off = 0;
  for (const CharClass& c : classes) {
    b.push_back(off);
    off += 1 + (int32_t)c.ranges.size();
  }
  b.push_back(off);
  b[6] = off;
  for (const CharClass& c : classes) {
    b.push_back(c.builtins);
    for (uint32_t r : c.ranges) b.push_back((int32_t)r);
  }
  return b;
}


namespace {
bool class_accepts(const CharClass& c, uint32_t ch, const uint8_t* flags) {
  for (size_t i = 0; i + 1 < c.ranges.size(); i += 2)
    if (ch >= c.ranges[i] && ch <= c.ranges[i + 1]) return true;
  if (!c.builtins || ch > 0xFFFF) return false;
  const unsigned f = flags[ch];  // callers pass ASCII only: code point == packed char
  const bool alnum = (f & 15) != 0;
  if ((c.builtins & BI_w) && (ch == '_' || alnum)) return true;
  if ((c.builtins & BI_s) && (f & 16)) return true;
  if ((c.builtins & BI_d) && (f & 4)) return true;
  if ((c.builtins & BI_W) && (ch != '\n' && ch != '_' && !alnum)) return true;
  if ((c.builtins & BI_S) && !(f & 16)) return true;
  if ((c.builtins & BI_D) && (ch != '\n' && !(f & 4))) return true;
  return false;
}
}  // namespace

std::vector<int32_t> Program::to_device_image(const uint8_t* flags) const {
  std::vector<int32_t> img = to_blob();
  // (the reference's starttype CHAR, regexec.inl:220-232: the search for the next start jumps to the program's first
  // character by length -- over NUL bytes; extras word 0 bit 1 tells the executors' skip to do the same)
  const bool char_first = start_inst >= 0 && (size_t)start_inst < insts.size() && insts[(size_t)start_inst].type == OP_CHAR;
  // (starttype BOL with `^`: the jump to the byte behind the next line feed, regexec.inl:233-246 -- extras word 0 bit 2)
  const bool bol_first = start_inst >= 0 && (size_t)start_inst < insts.size() && insts[(size_t)start_inst].type == OP_BOL && (uint32_t)insts[(size_t)start_inst].u1 == (uint32_t)'^';
  const size_t extra = img.size();
  img[7] = (int32_t)extra;
  img.resize(extra + 10 + 4 * classes.size(), 0);
  uint32_t* ex = reinterpret_cast<uint32_t*>(img.data() + extra);
  auto set = [](uint32_t* bm, unsigned c) { bm[c >> 5] |= 1u << (c & 31); };
  for (unsigned c = 0; c < 128; ++c)
    if (flags[c] & 15) set(ex + 6, c);
  for (size_t k = 0; k < classes.size(); ++k)
    for (unsigned c = 0; c < 128; ++c)
      if (class_accepts(classes[k], c, flags)) set(ex + 10 + 4 * k, c);
  // first-character prefilter: usable when every start thread reaches only
  // character-consuming instructions without passing a position test
  bool usable = !insts.empty();
  bool nonascii = false;
  uint32_t first[4] = {0, 0, 0, 0};
  std::vector<char> seen(insts.size(), 0);
  std::vector<int> todo;
  for (int32_t s : starts)
    if (s >= 0) todo.push_back(s);
  while (!todo.empty() && usable) {
    int id = todo.back();
    todo.pop_back();
    if (id < 0 || id >= (int)insts.size()) {
      usable = false;
      break;
    }
    if (seen[id]) continue;
    seen[id] = 1;
    const Inst& in = insts[id];
    switch (in.type) {
      case OP_OR:
        todo.push_back(in.u2);
        todo.push_back(in.u1);
        break;
      case OP_LBRA:
      case OP_RBRA:
        todo.push_back(in.u2);
        break;
      case OP_CHAR:
        if ((uint32_t)in.u1 < 128) set(first, (unsigned)in.u1);
        else nonascii = true;
        break;
      case OP_ANY:
        for (unsigned c = 0; c < 128; ++c)
          if (c != '\n') set(first, c);
        nonascii = true;
        break;
      case OP_ANYNL:
        for (unsigned c = 0; c < 128; ++c) set(first, c);
        nonascii = true;
        break;
      case OP_CCLASS:
      case OP_NCCLASS: {
        const bool neg = in.type == OP_NCCLASS;
        if (in.u1 < 0 || in.u1 >= (int)classes.size()) {
          usable = false;
          break;
        }
        for (unsigned c = 0; c < 128; ++c)
          if (class_accepts(classes[in.u1], c, flags) != neg) set(first, c);
        nonascii = true;  // conservative: any non-ASCII char stays a candidate
        break;
      }
      default:  // BOL EOL BOW NBOW END or an unknown opcode
        usable = false;
        break;
    }
  }
  ex[0] = (usable ? 1u : 0u) | (usable && char_first ? 2u : 0u) | (bol_first ? 4u : 0u);
  for (int k = 0; k < 4; ++k) ex[1 + k] = first[k];
  ex[5] = nonascii ? 1u : 0u;
  return img;
}

}  // namespace csrx
