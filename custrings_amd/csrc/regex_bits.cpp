// Host side of the bit-parallel regex form (regex_bits.h): which programs convert, and their image.
//
// The program (the reference's instruction stream, regcomp.cpp:314-1061 -- emitted word for word by regex_compile.cpp) is
// walked depth first from its start instruction, an OR's preferred branch (u1) first: the order in which the reference's
// executor activates threads (regexec.inl:69-91,204-442), hence the order of the alternatives' priorities.  Brackets are
// passed through (the bit form reports whole matches only).  Every path to END becomes one alternative.  A loop is
// accepted in one shape only: the greedy `+` over the LAST single-character item of the ONLY path (item; OR whose preferred
// branch returns to the item and whose other branch runs through brackets to END).
#include "regex_bits.h"

#include <algorithm>
#include <array>
#include <cstdlib>

#include "cs_config.h"
#include "regex_program.h"
#include "regex_vm.h"

namespace csrx {

namespace {

struct Item {
  int kind, arg, off;
  int inst = -1;  // (class items: the instruction they come from)
};
struct Alt {
  std::vector<Item> items;
  int len = 0;
  bool plus = false;
  std::vector<int> tail;  // assertions behind the trailing `+` loop (kinds)
};
using Members = std::array<uint32_t, 4>;  // ASCII member set of a consuming instruction

struct Walker {
  const Program& prog;
  const csvm::ProgView& V;
  std::vector<Alt> alts;
  std::vector<Members> classes;
  bool ok = true;
  int steps = 0;

  int class_of(const Members& m) {
    for (size_t k = 0; k < classes.size(); ++k)
      if (classes[k] == m) return (int)k;
    classes.push_back(m);
    return (int)classes.size() - 1;
  }
  bool consuming(int t) const { return t == OP_CHAR || t == OP_ANY || t == OP_ANYNL || t == OP_CCLASS || t == OP_NCCLASS; }
  Members members(const Inst& in) const {
    Members m{0, 0, 0, 0};
    for (unsigned c = 1; c < 128; ++c) {  // (a NUL byte never reaches the bit form: such sub-tiles take the automaton)
      bool hit = false;
      switch (in.type) {
        case OP_CHAR: hit = (uint32_t)in.u1 == c; break;
        case OP_ANY: hit = c != '\n'; break;
        case OP_ANYNL: hit = true; break;
        case OP_CCLASS: hit = csvm::class_match(V, in.u1, (csvm::Char)c); break;
        case OP_NCCLASS: hit = !csvm::class_match(V, in.u1, (csvm::Char)c); break;
      }
      if (hit) m[c >> 5] |= 1u << (c & 31);
    }
    return m;
  }
  int skip_brackets(int pc) const {
    size_t guard = 0;
    while (pc >= 0 && (size_t)pc < prog.insts.size() && (prog.insts[(size_t)pc].type == OP_LBRA || prog.insts[(size_t)pc].type == OP_RBRA) &&
           guard++ <= prog.insts.size())
      pc = prog.insts[(size_t)pc].u2;
    return pc;
  }
  void walk(int pc, Alt cur, std::vector<char>& on_path) {
    while (ok) {
      if (++steps > 4096 || pc < 0 || (size_t)pc >= prog.insts.size()) {
        ok = false;
        return;
      }
      if (on_path[(size_t)pc]) {  // a loop that is not the trailing `+`
        ok = false;
        return;
      }
      const Inst& in = prog.insts[(size_t)pc];
      if (in.type == OP_END) {
        if (cur.len < 1 || cur.len > csbits::kMaxLen || (int)alts.size() == csbits::kMaxAlts) {
          ok = false;
          return;
        }
        alts.push_back(cur);
        return;
      }
      if (in.type == OP_LBRA || in.type == OP_RBRA) {
        pc = in.u2;
        continue;
      }
      if ((int)cur.items.size() >= csbits::kMaxItems) {
        ok = false;
        return;
      }
      if (in.type == OP_OR) {
        on_path[(size_t)pc] = 1;
        walk(in.u1, cur, on_path);
        if (ok) walk(in.u2, cur, on_path);
        on_path[(size_t)pc] = 0;
        return;
      }
      if (in.type == OP_BOW || in.type == OP_NBOW) {
        cur.items.push_back({in.type == OP_BOW ? csbits::K_BOW : csbits::K_NBOW, 0, cur.len});
        pc = in.u2;
        continue;
      }
      if (in.type == OP_BOL) {
        cur.items.push_back({(uint32_t)in.u1 == (uint32_t)'^' ? csbits::K_BOL_MULTI : csbits::K_BOL, 0, cur.len});
        pc = in.u2;
        continue;
      }
      if (in.type == OP_EOL) {
        cur.items.push_back({(uint32_t)in.u1 == (uint32_t)'$' ? csbits::K_EOL_MULTI : csbits::K_EOL, 0, cur.len});
        pc = in.u2;
        continue;
      }
      if (!consuming(in.type)) {
        ok = false;
        return;
      }
      const int cls = class_of(members(in));
      cur.items.push_back({csbits::K_CLASS, cls, cur.len, pc});
      ++cur.len;
      // the trailing greedy `+`: item; OR(preferred -> the item, other -> brackets -> END)
      const int nx = skip_brackets(in.u2);
      if (nx >= 0 && (size_t)nx < prog.insts.size() && prog.insts[(size_t)nx].type == OP_OR) {
        const Inst& orr = prog.insts[(size_t)nx];
        if (skip_brackets(orr.u1) == pc) {
          int out = skip_brackets(orr.u2);
          // (brackets between the item and the OR would close and re-open a group inside the loop: only whole matches are
          // reported, so they do not matter)
          // behind the loop: END, or up to three assertions and END (the tail: `[^ ]+$`, `\w+\b`)
          while (out >= 0 && (size_t)out < prog.insts.size() && cur.tail.size() < 3) {
            const Inst& t = prog.insts[(size_t)out];
            int kind = -1;
            if (t.type == OP_BOW) kind = csbits::K_BOW;
            else if (t.type == OP_NBOW) kind = csbits::K_NBOW;
            else if (t.type == OP_BOL) kind = (uint32_t)t.u1 == (uint32_t)'^' ? csbits::K_BOL_MULTI : csbits::K_BOL;
            else if (t.type == OP_EOL) kind = (uint32_t)t.u1 == (uint32_t)'$' ? csbits::K_EOL_MULTI : csbits::K_EOL;
            if (kind < 0) break;
            cur.tail.push_back(kind);
            out = skip_brackets(t.u2);
          }
          if (cs::cfg("CS_NO_BITS_TAIL") && !cur.tail.empty()) out = -1;
          if (out < 0 || (size_t)out >= prog.insts.size() || prog.insts[(size_t)out].type != OP_END || !alts.empty()) {
            ok = false;
            return;
          }
          cur.plus = true;
          if (cur.len > csbits::kMaxLen) {
            ok = false;
            return;
          }
          alts.push_back(cur);
          return;
        }
      }
      pc = in.u2;
    }
  }
};

}  // namespace

std::vector<int32_t> build_bits(const Program& prog, const std::vector<int32_t>& image, const uint8_t* flags) {
  const std::vector<int32_t> none;
  if (prog.insts.empty() || cs::cfg("CS_NO_BITS")) return none;
  const csvm::ProgView V = csvm::make_view(image.data(), flags);
  Walker w{prog, V, {}, {}, true, 0};
  std::vector<char> on_path(prog.insts.size(), 0);
  w.walk(prog.start_inst, Alt{}, on_path);
  if (!w.ok || w.alts.empty()) return none;
  bool any_plus = false;
  for (const Alt& a : w.alts) any_plus = any_plus || a.plus;
  if (any_plus && w.alts.size() != 1) return none;
  // the classes the assertions need: word characters (regexec.inl: BOW / NBOW look at the alphanumeric flag, `_` is
  // not a word character there), the newline
  uint32_t fl = 0;
  for (const Alt& a : w.alts)
    for (const Item& it : a.items) {
      if (it.kind == csbits::K_BOW || it.kind == csbits::K_NBOW) fl |= csbits::F_WORD;
      if (it.kind == csbits::K_BOL) fl |= csbits::F_BOL;
      if (it.kind == csbits::K_EOL) fl |= csbits::F_EOL;
      if (it.kind == csbits::K_BOL_MULTI) fl |= csbits::F_BOL_MULTI;
      if (it.kind == csbits::K_EOL_MULTI) fl |= csbits::F_EOL_MULTI;
    }
  for (const Alt& a : w.alts)
    for (int kind : a.tail) {  // (the tail's masks are built where they are used, but the classes they read must exist)
      if (kind == csbits::K_BOW || kind == csbits::K_NBOW) fl |= csbits::F_WORD;
      if (kind == csbits::K_BOL_MULTI) fl |= csbits::F_BOL_MULTI;
      if (kind == csbits::K_EOL_MULTI) fl |= csbits::F_EOL_MULTI;
    }
  int word_cls = -1, nl_cls = -1;
  if (fl & csbits::F_WORD) {
    Members m{0, 0, 0, 0};
    for (unsigned c = 1; c < 128; ++c)
      if (csvm::is_word(V, (csvm::Char)c)) m[c >> 5] |= 1u << (c & 31);
    word_cls = w.class_of(m);
  }
  if (fl & (csbits::F_BOL_MULTI | csbits::F_EOL_MULTI)) {
    Members m{0, 0, 0, 0};
    m[0] = 1u << '\n';
    nl_cls = w.class_of(m);
  }
  if ((int)w.classes.size() > csbits::kMaxClasses) return none;
  const Alt& first = w.alts[0];
  int plus_cls = -1;
  if (first.plus) {
    plus_cls = first.items.back().arg;
    fl |= csbits::F_PLUS;
    if (first.items.size() == 1 && first.tail.empty()) fl |= csbits::F_PURE_PLUS;
    if (!first.tail.empty()) fl |= csbits::F_TAIL;
    if (!first.tail.empty() && first.items.size() == 1 && !cs::cfg("CS_NO_BITS_PURE_TAIL")) fl |= csbits::F_PURE_TAIL;
  }
  bool same = true;
  for (const Alt& a : w.alts) same = same && a.len == first.len;
  if (same) fl |= csbits::F_SAME_LEN;
  if (w.alts.size() == 1 && first.items.size() == 1 && first.items[0].kind == csbits::K_CLASS && first.tail.empty()) {
    const Inst& in = prog.insts[(size_t)first.items[0].inst];
    bool bytewise = false, high = false;
    if (in.type == OP_CHAR) {
      bytewise = (uint32_t)in.u1 >= 1u && (uint32_t)in.u1 < 128u;
    } else if (in.type == OP_ANY || in.type == OP_ANYNL) {
      bytewise = high = true;
    } else if ((in.type == OP_CCLASS || in.type == OP_NCCLASS) && in.u1 >= 0 && (size_t)in.u1 < prog.classes.size()) {
      const CharClass& cc = prog.classes[(size_t)in.u1];
      bytewise = cc.builtins == 0;
      for (uint32_t r : cc.ranges) bytewise = bytewise && r < 128u;
      high = in.type == OP_NCCLASS;
    }
    if (bytewise) fl |= csbits::F_BYTE_CLASS | (high ? csbits::F_HIGH_MEMBER : 0u);
    if (in.type == OP_CHAR && prog.start_inst == first.items[0].inst) fl |= csbits::F_CHAR_FIRST;
    if (!bytewise && (in.type == OP_CCLASS || in.type == OP_NCCLASS) && in.u1 >= 0 && (size_t)in.u1 < prog.classes.size()) {
      const CharClass& cc = prog.classes[(size_t)in.u1];
      bool ascii_ranges = cc.builtins != 0 && (cc.builtins & ~63) == 0;
      for (uint32_t r : cc.ranges) ascii_ranges = ascii_ranges && r < 128u;
      if (ascii_ranges && !cs::cfg("CS_NO_FLAG_CLASS")) fl |= csbits::F_FLAG_CLASS | ((uint32_t)cc.builtins << 16) | (in.type == OP_NCCLASS ? 1u << 22 : 0u);
    }
  }

  std::vector<int32_t> img(csbits::kHeaderWords + csbits::kTableWords, 0);
  img[0] = csbits::kMagic;
  img[1] = (int32_t)w.classes.size();
  img[2] = (int32_t)fl;
  img[3] = (int32_t)w.alts.size();
  img[5] = plus_cls;
  img[6] = word_cls;
  img[7] = nl_cls;
  for (unsigned c = 0; c < 128; ++c) {
    uint32_t set = 0;
    for (size_t k = 0; k < w.classes.size(); ++k)
      if ((w.classes[k][c >> 5] >> (c & 31)) & 1u) set |= 1u << k;
    img[csbits::kHeaderWords + (c >> 2)] |= (int32_t)(set << (8 * (c & 3)));
  }
  for (const Alt& a : w.alts) {
    img.push_back((int32_t)(a.items.size() | ((size_t)a.len << 8)));
    for (const Item& it : a.items) img.push_back(it.kind | (it.arg << 8) | (it.off << 16));
  }
  if (!first.tail.empty()) {
    uint32_t tw = (uint32_t)first.tail.size();
    for (size_t i = 0; i < first.tail.size(); ++i) tw |= (uint32_t)first.tail[i] << (8 * (i + 1));
    img.push_back((int32_t)tw);
  }
  img[4] = (int32_t)img.size();
  return img;
}

}  // namespace csrx
