// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference algorithms on the hot path of
// rapidsai/custrings, used only as the parity checker by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
// custrings_amd/ may include, link or call this file.
//
// It deliberately follows the reference's own formulation (character
// positions, the custring_view helper semantics, the list-based NFA
// simulator) rather than the product's byte-offset/bitmask formulation, so
// that comparing the two is a genuine differential test.  Every function
// cites the reference file:line it restates (paths under /root/reference).
//
// Pinning: the reference's device code cannot be built in this image (it
// needs CUDA, Thrust and RMM; no stand-ins are written for them).  The oracle
// is pinned instead against (i) the known answers held by the reference's own
// gtests / pytest files (tests/golden/reference_tests.json), (ii) the vectors
// recorded from the reference in SURVEY.md Appendix A
// (tests/golden/survey_appendix_a.json), (iii) pandas-generated expectations
// for the pandas-compared reference tests, and (iv) the real reference regex
// compiler built in place into oracle/_ref (regcomp.cpp is pure host C++).
//
// Strings are Arrow-style: chars + int64 offsets + optional validity bitmask
// (LSB first, 1 = valid).  Valid UTF-8 input is assumed, as in the reference.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "unicode_tables_gen.h"  // generated: orc_unicode_flags[65536], orc_charcases[65536]

typedef uint32_t Char;  // 1-4 raw UTF-8 bytes packed big-endian (custring_view.inl:1724-1744)

// ------------------------------------------------------------------ column --
struct orc_col {
  int64_t rows = 0;
  std::vector<int64_t> off;    // rows+1
  std::vector<uint8_t> chars;
  std::vector<uint8_t> valid;  // empty = all valid; else (rows+7)/8 bytes
  bool is_valid(int64_t r) const { return valid.empty() || ((valid[r >> 3] >> (r & 7)) & 1); }
};

struct View {  // one non-null row, the custring_view equivalent
  const uint8_t* d;
  unsigned bytes;
  unsigned nchars;
};

// custring_view.inl:48-57
static inline unsigned lead_width(uint8_t b) {
  unsigned n = 1;
  n += (b & 0xF0) == 0xF0;
  n += (b & 0xE0) == 0xE0;
  n += (b & 0xC0) == 0xC0;
  n -= (b & 0xC0) == 0x80;
  return n;
}
// custring_view.inl:1758-1766
static inline unsigned count_chars(const uint8_t* s, unsigned bytes) {
  unsigned n = 0;
  for (unsigned i = 0; i < bytes; ++i) n += (s[i] & 0xC0) != 0x80;
  return n;
}
// custring_view.inl:1724-1744; bytes past `end` read as the NUL terminator
static inline unsigned decode(const uint8_t* p, const uint8_t* end, Char& c) {
  unsigned w = (p < end) ? lead_width(*p) : 1;
  c = (p < end) ? *p : 0;
  for (unsigned k = 1; k < w; ++k) c = (c << 8) | ((p + k < end) ? p[k] : 0);
  return w;
}
// custring_view.inl:1714-1722
static inline unsigned packed_width(Char c) {
  return 1 + ((c & 0xFF00u) > 0) + ((c & 0xFF0000u) > 0) + ((c & 0xFF000000u) > 0);
}
// custring_view.inl:1746-1755
static inline void append_packed(std::vector<uint8_t>& out, Char c) {
  unsigned w = packed_width(c);
  for (unsigned k = 0; k < w; ++k) out.push_back((uint8_t)(c >> (8 * (w - 1 - k))));
}
// util.inl:22-75
static inline unsigned cp_to_packed(unsigned u) {
  if (u < 0x80) return u;
  if (u < 0x800) return ((u << 2) & 0x1F00) | (u & 0x3F) | 0xC080;
  if (u < 0x10000) return ((u << 4) & 0x0F0000) | ((u << 2) & 0x003F00) | (u & 0x3F) | 0xE08080;
  if (u < 0x110000)
    return ((u << 6) & 0x07000000) | ((u << 4) & 0x003F0000) | ((u << 2) & 0x3F00) | (u & 0x3F) |
           0xF0808080u;
  return 0;
}
static inline unsigned packed_to_cp(unsigned c) {
  if (c < 0x80) return c;
  if (c < 0xE000) return ((c & 0x1F00) >> 2) | (c & 0x3F);
  if (c < 0xF00000) return ((c & 0x0F0000) >> 4) | ((c & 0x3F00) >> 2) | (c & 0x3F);
  if (c <= 0xF8000000u)  // note the 0x03 mask on the lead byte (util.inl:67)
    return ((c & 0x03000000) >> 6) | ((c & 0x3F0000) >> 4) | ((c & 0x3F00) >> 2) | (c & 0x3F);
  return 0;
}
static inline View make_view(const orc_col& c, int64_t r) {
  View v;
  v.d = c.chars.data() + c.off[r];
  v.bytes = (unsigned)(c.off[r + 1] - c.off[r]);
  v.nchars = count_chars(v.d, v.bytes);
  return v;
}
// custring_view.inl:261-281 (lengths come from lead bytes, like the 2-bit table)
static unsigned byte_pos(const View& v, unsigned chpos) {
  if (chpos == 0) return 0;
  if (chpos >= v.nchars) return v.bytes;
  if (v.nchars == v.bytes) return chpos;
  unsigned off = 0, seen = 0;
  for (unsigned i = 0; i < v.bytes && seen < chpos; ++i) {
    unsigned w = lead_width(v.d[i]);
    if (w) {
      off += w;
      ++seen;
    }
  }
  return off;
}
// custring_view.inl:1793-1797
static inline unsigned char_pos(const View& v, unsigned bytepos) {
  return v.bytes == v.nchars ? bytepos : count_chars(v.d, bytepos);
}
static inline Char char_at(const View& v, unsigned chpos) {
  Char c;
  decode(v.d + byte_pos(v, chpos), v.d + v.bytes, c);
  return c;
}

// custring_view.inl:481-514: naive byte search, char positions in and out
static int find_str(const View& v, const uint8_t* s, unsigned sbytes, unsigned pos = 0, int count = -1) {
  if (!s || !sbytes) return -1;
  int nchars = (int)v.nchars;
  if (count < 0) count = nchars;
  int end = (int)pos + count;
  if (end < 0 || end > nchars) end = nchars;
  int spos = (int)byte_pos(v, pos);
  int epos = (int)byte_pos(v, (unsigned)end);
  int span = (epos - spos) - (int)sbytes + 1;
  for (int i = 0; i < span; ++i)
    if (memcmp(v.d + spos + i, s, sbytes) == 0) return (int)char_pos(v, (unsigned)(i + spos));
  return -1;
}
// custring_view.inl:517-543
static int find_char(const View& v, Char ch, unsigned pos = 0, int count = -1) {
  int nchars = (int)v.nchars;
  if (count < 0) count = nchars;
  int end = (int)pos + count;
  if (end < 0 || end > nchars) end = nchars;
  if ((int)pos > end || ch == 0 || v.bytes == 0) return -1;
  int spos = (int)byte_pos(v, pos);
  int epos = (int)byte_pos(v, (unsigned)end);
  int last = (epos - spos) - (int)packed_width(ch);
  for (int i = 0; i <= last; ++i) {
    Char c;
    const uint8_t* p = v.d + spos + i;
    // decode at every byte offset; a continuation byte decodes to itself
    unsigned w = lead_width(*p);
    c = *p;
    for (unsigned k = 1; k < w; ++k) c = (c << 8) | ((p + k < v.d + v.bytes) ? p[k] : 0);
    if (c == ch) return (int)count_chars(v.d, (unsigned)(i + spos));
  }
  return -1;
}
// custring_view.inl:93-105
static bool in_set(const uint8_t* set, Char ch) {
  const uint8_t* end = set + strlen((const char*)set);
  Char t;
  unsigned w = decode(set, end, t);
  while (t) {
    if (t == ch) return true;
    set += w;
    w = decode(set, end, t);
  }
  return false;
}

// ------------------------------------------------------- output assembly --
struct Builder {
  std::vector<int64_t> off{0};
  std::vector<uint8_t> chars;
  std::vector<uint8_t> nulls;  // byte per row, 1 = null
  void add_null() {
    off.push_back((int64_t)chars.size());
    nulls.push_back(1);
  }
  void add(const uint8_t* p, size_t n) {
    chars.insert(chars.end(), p, p + n);
    off.push_back((int64_t)chars.size());
    nulls.push_back(0);
  }
  void close_row() {  // row bytes were appended to chars directly
    off.push_back((int64_t)chars.size());
    nulls.push_back(0);
  }
  orc_col* finish() {
    orc_col* c = new orc_col;
    c->rows = (int64_t)nulls.size();
    c->off.swap(off);
    c->chars.swap(chars);
    bool any = false;
    for (uint8_t n : nulls) any |= (n != 0);
    if (any) {
      c->valid.assign((c->rows + 7) / 8, 0);
      for (int64_t r = 0; r < c->rows; ++r)
        if (!nulls[r]) c->valid[r >> 3] |= (uint8_t)(1u << (r & 7));
    }
    return c;
  }
};

// ------------------------------------------------------------------ regex --
// Program image: custrings_amd/csrc/regex_program.h layout (int32 words).
struct Prog {
  const int32_t* w;
  int start, ninst, nstarts, nclasses;
  const int32_t *insts, *starts, *cls_off, *cls_data;
  explicit Prog(const int32_t* words) : w(words) {
    start = w[1];
    ninst = w[3];
    nstarts = w[4];
    nclasses = w[5];
    insts = w + 8;
    starts = insts + 4 * ninst;
    cls_off = starts + nstarts;
    cls_data = cls_off + nclasses + 1;
  }
  int type(int i) const { return insts[4 * i]; }
  int u1(int i) const { return insts[4 * i + 1]; }
  int u2(int i) const { return insts[4 * i + 2]; }
};
enum {
  CHAR = 0177, RBRA = 0201, LBRA = 0202, OR = 0204, ANY = 0300, ANYNL = 0301, BOL = 0303,
  EOL = 0304, CCLASS = 0305, NCCLASS = 0306, BOW = 0307, NBOW = 0310, END = 0377
};
static inline bool fl_alnum(uint8_t f) { return (f & 15) != 0; }
// regexec.inl:127-155
static bool class_match(const Prog& p, int cls, Char ch) {
  const int32_t* c = p.cls_data + p.cls_off[cls];
  int n = p.cls_off[cls + 1] - p.cls_off[cls] - 1;
  int builtins = c[0];
  for (int i = 0; i < n; i += 2)
    if (ch >= (Char)c[1 + i] && ch <= (Char)c[2 + i]) return true;
  if (!builtins) return false;
  unsigned u = packed_to_cp(ch);
  if (u > 0xFFFF) return false;
  uint8_t f = orc_unicode_flags[u];
  if ((builtins & 1) && (ch == '_' || fl_alnum(f))) return true;
  if ((builtins & 2) && (f & 16)) return true;
  if ((builtins & 4) && (f & 4)) return true;
  if ((builtins & 8) && (ch != '\n' && ch != '_' && !fl_alnum(f))) return true;
  if ((builtins & 16) && !(f & 16)) return true;
  if ((builtins & 32) && (ch != '\n' && !(f & 4))) return true;
  return false;
}

struct ThreadList {  // regexec.inl:26-108: ordered, first activation of an inst wins
  std::vector<int> ids, bx, by;
  std::vector<uint8_t> seen;
  explicit ThreadList(int n) : seen(n, 0) {}
  void reset() {
    for (int i : ids) seen[i] = 0;
    ids.clear();
    bx.clear();
    by.clear();
  }
  void activate(int id, int x, int y) {
    if (seen[id]) return;
    seen[id] = 1;
    ids.push_back(id);
    bx.push_back(x);
    by.push_back(y);
  }
};

static bool is_word(Char c) {
  unsigned u = packed_to_cp(c);
  return u < 0x10000 && fl_alnum(orc_unicode_flags[u]);
}

// regexec.inl:204-442.  [begin,end) on entry is the window of allowed START
// positions; on success they become the match span (group id 0) or the range
// the highest-priority thread recorded for capture group `group_id`
// (regexec.inl:268,296-307,425-428; -1 where that group took no part).
static int nfa_run(const Prog& p, const View& v, int& begin, int& end, int group_id = 0) {
  int match = 0;
  int first_type = p.type(p.start);
  int fast = (first_type == CHAR || first_type == BOL) ? first_type : 0;
  Char fast_ch = fast ? (Char)p.u1(p.start) : 0;
  bool check_start = fast != 0;
  int txtlen = (int)v.nchars;
  int pos = begin, eos = end;
  Char c = 0;
  ThreadList a(p.ninst), b(p.ninst);
  ThreadList *cur = &a, *nxt = &b;
  int guard_limit = 4 * p.ninst + 8;
  do {
    if (check_start) {
      if (fast == CHAR) {
        int f = find_char(v, fast_ch, (unsigned)pos);
        if (f < 0) return match;
        pos = f;
      } else if (fast == BOL && pos != 0) {
        if (fast_ch != '^') return match;
        --pos;
        int f = find_char(v, (Char)'\n', (unsigned)pos);
        if (f < 0) return match;
        pos = f + 1;
      }
    }
    if ((eos < 0 || pos < eos) && match == 0)
      for (int i = 0; p.starts[i] >= 0; ++i) cur->activate(p.starts[i], group_id == 0 ? pos : -1, -1);
    c = pos >= txtlen ? 0 : char_at(v, (unsigned)pos);
    // expand the non-consuming instructions until a fixed point
    bool expanded;
    int guard = 0;
    do {
      nxt->reset();
      expanded = false;
      for (size_t i = 0; i < cur->ids.size(); ++i) {
        int id = cur->ids[i], x = cur->bx[i], y = cur->by[i];
        int go = -1;
        switch (p.type(id)) {
          case CHAR: case ANY: case ANYNL: case CCLASS: case NCCLASS: case END:
            go = id;
            break;
          case LBRA:
            if (p.u1(id) == group_id) x = pos;
            go = p.u2(id);
            expanded = true;
            break;
          case RBRA:
            if (p.u1(id) == group_id) y = pos;
            go = p.u2(id);
            expanded = true;
            break;
          case BOL:
            if (pos == 0 || ((Char)p.u1(id) == '^' && char_at(v, (unsigned)pos - 1) == '\n')) {
              go = p.u2(id);
              expanded = true;
            }
            break;
          case EOL:
            if (c == 0 || ((Char)p.u1(id) == '$' && c == '\n')) {
              go = p.u2(id);
              expanded = true;
            }
            break;
          case BOW: case NBOW: {
            bool cw = is_word(c);
            bool lw = is_word(pos ? char_at(v, (unsigned)pos - 1) : 0);
            if ((cw != lw) == (p.type(id) == BOW)) {
              go = p.u2(id);
              expanded = true;
            }
            break;
          }
          case OR:
            nxt->activate(p.u1(id), x, y);
            go = p.u2(id);
            expanded = true;
            break;
        }
        if (go >= 0) nxt->activate(go, x, y);
      }
      std::swap(cur, nxt);
      // the reference spins forever on an empty-loop pattern such as (a*)*;
      // after ninst passes every reachable instruction has been activated
    } while (expanded && ++guard < guard_limit);
    // consume c
    nxt->reset();
    for (size_t i = 0; i < cur->ids.size(); ++i) {
      int id = cur->ids[i], x = cur->bx[i], y = cur->by[i];
      int go = -1;
      bool stop = false;
      switch (p.type(id)) {
        case CHAR: if ((Char)p.u1(id) == c) go = p.u2(id); break;
        case ANY: if (c != '\n') go = p.u2(id); break;
        case ANYNL: go = p.u2(id); break;
        case CCLASS: if (class_match(p, p.u1(id), c)) go = p.u2(id); break;
        case NCCLASS: if (!class_match(p, p.u1(id), c)) go = p.u2(id); break;
        case END:
          match = 1;
          begin = x;
          end = group_id == 0 ? pos : y;
          stop = true;  // lower-priority threads are cut off
          break;
      }
      if (stop) break;
      if (go >= 0) nxt->activate(go, x, y);
    }
    ++pos;
    std::swap(cur, nxt);
    check_start = fast && cur->ids.empty();
  } while (c && (!cur->ids.empty() || match == 0));
  return match;
}
// regexec.inl:456-463
static int re_find(const Prog& p, const View& v, int& begin, int& end) {
  int r = nfa_run(p, v, begin, end);
  if (r <= 0) begin = end = -1;
  return r;
}

// ------------------------------------------------------------------ C API --
extern "C" {

orc_col* orc_col_create(int64_t rows, const int64_t* off, const uint8_t* chars, const uint8_t* valid) {
  orc_col* c = new orc_col;
  c->rows = rows;
  c->off.assign(off, off + rows + 1);
  c->chars.assign(chars, chars + off[rows]);
  if (valid) c->valid.assign(valid, valid + (rows + 7) / 8);
  return c;
}
void orc_col_free(orc_col* c) { delete c; }
int64_t orc_col_rows(const orc_col* c) { return c->rows; }
int64_t orc_col_nbytes(const orc_col* c) { return (int64_t)c->chars.size(); }
const int64_t* orc_col_offsets(const orc_col* c) { return c->off.data(); }
const uint8_t* orc_col_chars(const orc_col* c) { return c->chars.data(); }
// always materialised: (rows+7)/8 bytes, LSB first, 1 = valid (NVStrings.cu:493-544)
void orc_col_bitmask(const orc_col* c, uint8_t* out) {
  int64_t nb = (c->rows + 7) / 8;
  memset(out, 0, (size_t)nb);
  for (int64_t r = 0; r < c->rows; ++r)
    if (c->is_valid(r)) out[r >> 3] |= (uint8_t)(1u << (r & 7));
}
int64_t orc_col_null_count(const orc_col* c) {
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) n += !c->is_valid(r);
  return n;
}
// attrs.cu:72-112 byte_count: -1 for null
void orc_byte_count(const orc_col* c, int32_t* out) {
  for (int64_t r = 0; r < c->rows; ++r)
    out[r] = c->is_valid(r) ? (int32_t)(c->off[r + 1] - c->off[r]) : -1;
}

// case.cu:31-97 (lower), :100-170 (upper)
static orc_col* change_case(const orc_col* c, unsigned flag_bit) {
  Builder b;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) {
      b.add_null();
      continue;
    }
    View v = make_view(*c, r);
    const uint8_t *p = v.d, *e = v.d + v.bytes;
    for (unsigned k = 0; k < v.nchars; ++k) {
      Char ch;
      unsigned w = decode(p, e, ch);
      p += w;
      unsigned u = packed_to_cp(ch);
      unsigned f = u <= 0xFFFF ? orc_unicode_flags[u] : 0;
      if (f & flag_bit) ch = cp_to_packed(orc_charcases[u]);
      append_packed(b.chars, ch);
    }
    b.close_row();
  }
  return b.finish();
}
orc_col* orc_lower(const orc_col* c) { return change_case(c, 32); }
orc_col* orc_upper(const orc_col* c) { return change_case(c, 64); }

// strip.cu:30-199 -> custring_view.inl:1398-1598. side 0 both, 1 left, 2 right
orc_col* orc_strip(const orc_col* c, const char* to_strip, int side) {
  const uint8_t* set = (const uint8_t*)(to_strip ? to_strip : " \n\t");
  Builder b;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) {
      b.add_null();
      continue;
    }
    View v = make_view(*c, r);
    const uint8_t* e = v.d + v.bytes;
    unsigned lead = 0, trail = 0;
    if (side != 2) {
      const uint8_t* p = v.d;
      for (unsigned k = 0; k < v.nchars; ++k) {
        Char ch;
        unsigned w = decode(p, e, ch);
        if (!in_set(set, ch)) break;
        p += w;
        lead += w;
      }
    }
    if (lead == v.bytes) {  // everything stripped (also the empty string)
      b.add(v.d, 0);
      continue;
    }
    if (side != 1) {
      const uint8_t* p = e;
      for (unsigned k = 0; k < v.nchars; ++k) {
        do --p; while (lead_width(*p) == 0);
        Char ch;
        unsigned w = decode(p, e, ch);
        if (!in_set(set, ch)) break;
        trail += w;
      }
    }
    if (lead + trail > v.bytes) {
      b.add(v.d, 0);
      continue;
    }
    b.add(v.d + lead, v.bytes - lead - trail);
  }
  return b.finish();
}

// find.cu:75-120. returns count of results != -1 (null rows give -2 and count)
int64_t orc_find(const orc_col* c, const char* str, int start, int end, int32_t* out) {
  unsigned bytes = (unsigned)strlen(str);
  if (start < 0) start = 0;
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r))
      out[r] = -2;
    else
      out[r] = find_str(make_view(*c, r), (const uint8_t*)str, bytes, (unsigned)start, end - start);
    n += out[r] != -1;
  }
  return n;
}
// find.cu:237-272
int64_t orc_contains(const orc_col* c, const char* str, uint8_t* out) {
  unsigned bytes = (unsigned)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->is_valid(r) && find_str(make_view(*c, r), (const uint8_t*)str, bytes) >= 0;
    n += out[r];
  }
  return n;
}
// modify.cu:109-192
orc_col* orc_replace(const orc_col* c, const char* str, const char* repl, int maxrepl) {
  if (!str || !*str) return nullptr;  // reference throws std::invalid_argument
  if (!repl) repl = "";
  unsigned sb = (unsigned)strlen(str), rb = (unsigned)strlen(repl);
  unsigned sc = count_chars((const uint8_t*)str, sb);
  Builder b;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) {
      b.add_null();
      continue;
    }
    View v = make_view(*c, r);
    int left = maxrepl < 0 ? (int)v.nchars : maxrepl;
    int pos = find_str(v, (const uint8_t*)str, sb);
    unsigned copied = 0;
    while (pos >= 0 && left > 0) {
      unsigned at = byte_pos(v, (unsigned)pos);
      b.chars.insert(b.chars.end(), v.d + copied, v.d + at);
      b.chars.insert(b.chars.end(), repl, repl + rb);
      copied = at + sb;
      pos = find_str(v, (const uint8_t*)str, sb, (unsigned)pos + sc);
      --left;
    }
    b.chars.insert(b.chars.end(), v.d + copied, v.d + v.bytes);
    b.close_row();
  }
  return b.finish();
}

// split.cu:32-48 + custring_view.inl:1223-1250 (delimiter token count)
static int count_delim_tokens(const View& v, const uint8_t* d, unsigned dbytes, int tokens) {
  if (v.bytes == 0) return 1;
  unsigned n = 0;
  int pos = find_str(v, d, dbytes);
  while (pos >= 0) {
    ++n;
    pos = find_str(v, d, dbytes, (unsigned)pos + dbytes);  // advances by delimiter BYTES
  }
  unsigned total = n + 1;
  if (tokens > 0 && total > (unsigned)tokens) total = (unsigned)tokens;
  return (int)total;
}
// split.cu:52-87
static int count_ws_tokens(const View& v, int tokens) {
  int n = 0;
  bool spaces = true;
  const uint8_t *p = v.d, *e = v.d + v.bytes;
  unsigned k = 0;
  while (k < v.nchars) {
    Char ch;
    unsigned w = decode(p, e, ch);
    if (spaces == (ch <= ' ')) {
      p += w;
      ++k;
    } else {
      n += (int)spaces;
      spaces = !spaces;
    }
  }
  if (tokens && n > tokens) n = tokens;
  if (n == 0) n = 1;
  return n;
}
// split.cu:734-822 (delimiter) and :863-956 (whitespace); column-major output.
// Returns the number of columns; cols receives malloc'd array of handles.
int orc_split(const orc_col* c, const char* delim, int maxsplit, orc_col*** cols_out) {
  int tokens = maxsplit > 0 ? maxsplit + 1 : 0;
  unsigned dbytes = delim ? (unsigned)strlen(delim) : 0;
  int dchars = delim ? (int)count_chars((const uint8_t*)delim, dbytes) : 0;
  std::vector<int> counts(c->rows, 0);
  int ncols = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) continue;
    View v = make_view(*c, r);
    counts[r] = delim ? count_delim_tokens(v, (const uint8_t*)delim, dbytes, tokens)
                      : count_ws_tokens(v, tokens);
    ncols = std::max(ncols, counts[r]);
  }
  int nout = ncols ? ncols : 1;  // no columns -> one all-null column (split.cu:756-757)
  orc_col** cols = (orc_col**)malloc(sizeof(orc_col*) * nout);
  for (int col = 0; col < nout; ++col) {
    Builder b;
    for (int64_t r = 0; r < c->rows; ++r) {
      if (!c->is_valid(r) || col >= counts[r]) {
        b.add_null();
        continue;
      }
      View v = make_view(*c, r);
      int nchars = (int)v.nchars, spos = 0, epos = nchars;
      if (delim) {
        for (int k = 0; k < counts[r] - 1; ++k) {
          epos = find_str(v, (const uint8_t*)delim, dbytes, (unsigned)spos);
          if (epos < 0) {
            epos = nchars;
            break;
          }
          if (k == col) break;
          spos = epos + dchars;  // advances by delimiter CHARS
          epos = nchars;
        }
        if (spos < epos) {
          unsigned s = byte_pos(v, (unsigned)spos), e = byte_pos(v, (unsigned)epos);
          b.add(v.d + s, e - s);
        } else {
          b.add(v.d, 0);  // empty, not null
        }
      } else {
        int k = 0;
        bool spaces = true;
        for (int pos = 0; pos < nchars; ++pos) {
          Char ch = char_at(v, (unsigned)pos);
          if (spaces == (ch <= ' ')) {
            if (spaces) spos = pos + 1;
            else epos = pos + 1;
            continue;
          }
          if (!spaces) {
            epos = nchars;
            if ((k + 1) == tokens) break;
            epos = pos;
            if (k == col) break;
            spos = pos + 1;
            epos = nchars;
            ++k;
          }
          spaces = !spaces;
        }
        if (spos < epos) {
          unsigned s = byte_pos(v, (unsigned)spos), e = byte_pos(v, (unsigned)epos);
          b.add(v.d + s, e - s);
        } else {
          b.add_null();  // whitespace split never yields empty strings
        }
      }
    }
    cols[col] = b.finish();
  }
  *cols_out = cols;
  return nout;
}
// custring_view.inl:550-582 rfind(str, bytes, pos, count): last occurrence lying inside the
// character window [pos, pos+count); -1 when there is none
static int rfind_str(const View& v, const uint8_t* s, unsigned sbytes, unsigned pos, int count) {
  if (!s || !sbytes) return -1;
  int nchars = (int)v.nchars;
  int end = (int)pos + count;
  if (end < 0 || end > nchars) end = nchars;
  int spos = (int)byte_pos(v, pos);
  int epos = (int)byte_pos(v, (unsigned)end);
  int span = (epos - spos) - (int)sbytes + 1;
  for (int i = 0; i < span; ++i)
    if (memcmp(v.d + epos - (int)sbytes - i, s, sbytes) == 0) return (int)char_pos(v, (unsigned)(epos - (int)sbytes - i));
  return -1;
}
// split.cu:960-1053 (delimiter) and :1055-1148 (whitespace): column-major rsplit.  The token
// COUNT is the forward one of split (same token_counter); the tokens are located from the right.
int orc_rsplit(const orc_col* c, const char* delim, int maxsplit, orc_col*** cols_out) {
  int tokens = maxsplit > 0 ? maxsplit + 1 : 0;
  unsigned dbytes = delim ? (unsigned)strlen(delim) : 0;
  int dchars = delim ? (int)count_chars((const uint8_t*)delim, dbytes) : 0;
  std::vector<int> counts(c->rows, 0);
  int ncols = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) continue;
    View v = make_view(*c, r);
    counts[r] = delim ? count_delim_tokens(v, (const uint8_t*)delim, dbytes, tokens) : count_ws_tokens(v, tokens);
    ncols = std::max(ncols, counts[r]);
  }
  int nout = ncols ? ncols : 1;  // split.cu:980-982
  orc_col** cols = (orc_col**)malloc(sizeof(orc_col*) * nout);
  for (int col = 0; col < nout; ++col) {
    Builder b;
    for (int64_t r = 0; r < c->rows; ++r) {
      if (!c->is_valid(r) || col >= counts[r]) {
        b.add_null();
        continue;
      }
      View v = make_view(*c, r);
      int dcount = counts[r];
      int nchars = (int)v.nchars, spos = 0, epos = nchars;
      if (delim) {
        for (int k = dcount - 1; k > 0; --k) {  // split.cu:1006-1021
          spos = rfind_str(v, (const uint8_t*)delim, dbytes, 0, epos);
          if (spos < 0) {
            spos = 0;
            break;
          }
          if (k == col) {
            spos += dchars;
            break;
          }
          epos = spos;
          spos = 0;
        }
        if (spos < epos) {
          unsigned s = byte_pos(v, (unsigned)spos), e = byte_pos(v, (unsigned)epos);
          b.add(v.d + s, e - s);
        } else {
          b.add(v.d, 0);  // empty, not null (split.cu:1031-1034)
        }
      } else {
        int k = dcount - 1;  // split.cu:1098-1124
        bool spaces = true;
        for (int pos = nchars; pos > 0; --pos) {
          Char ch = char_at(v, (unsigned)pos - 1);
          if (spaces == (ch <= ' ')) {
            if (spaces) epos = pos - 1;
            else spos = pos - 1;
            continue;
          }
          if (!spaces) {
            spos = 0;
            if ((ncols - k) == tokens) break;  // (the COLUMN count of the whole call, as the reference has it)
            spos = pos;
            if (k == col) break;
            epos = pos - 1;
            spos = 0;
            --k;
          }
          spaces = !spaces;
        }
        if (spos < epos) {
          unsigned s = byte_pos(v, (unsigned)spos), e = byte_pos(v, (unsigned)epos);
          b.add(v.d + s, e - s);
        } else {
          b.add_null();
        }
      }
    }
    cols[col] = b.finish();
  }
  *cols_out = cols;
  return nout;
}
void orc_free(void* p) { free(p); }

// count.cu:36-56,59-110: mode 0 contains_re, 1 match (start window = [0,1))
int64_t orc_contains_re(const orc_col* c, const int32_t* prog, int mode, uint8_t* out) {
  Prog p(prog);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = 0;
    if (c->is_valid(r)) {
      View v = make_view(*c, r);
      int b = 0, e = mode ? 1 : (int)v.nchars;
      out[r] = re_find(p, v, b, e) > 0;
    }
    n += out[r];
  }
  return n;
}
// count.cu:168-250 count_re: successive non-overlapping finds
int64_t orc_count_re(const orc_col* c, const int32_t* prog, int32_t* out) {
  Prog p(prog);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = 0;  // null rows count 0 (count.cu:179)
    if (!c->is_valid(r)) continue;
    View v = make_view(*c, r);
    int nchars = (int)v.nchars, b = 0, e = nchars, k = 0;
    while (b <= nchars && re_find(p, v, b, e) > 0) {
      ++k;
      b = e > b ? e : e + 1;
      e = nchars;
    }
    out[r] = k;
    n += k > 0;
  }
  return n;
}
// replace.cu:39-107,110-189
orc_col* orc_replace_re(const orc_col* c, const int32_t* prog, const char* repl, int maxrepl) {
  Prog p(prog);
  if (!repl) repl = "";
  size_t rb = strlen(repl);
  Builder b;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) {
      b.add_null();
      continue;
    }
    View v = make_view(*c, r);
    int left = maxrepl < 0 ? (int)v.nchars : maxrepl;
    int begin = 0, end = (int)v.nchars;
    unsigned copied = 0;
    while (left > 0) {
      if (re_find(p, v, begin, end) <= 0) break;
      unsigned s = byte_pos(v, (unsigned)begin);
      b.chars.insert(b.chars.end(), v.d + copied, v.d + s);
      b.chars.insert(b.chars.end(), repl, repl + rb);
      copied = byte_pos(v, (unsigned)end);
      begin = end;  // a zero-length match does not advance (replace.cu:91-93)
      end = (int)v.nchars;
      --left;
    }
    b.chars.insert(b.chars.end(), v.d + copied, v.d + v.bytes);
    b.close_row();
  }
  return b.finish();
}

// extract.cu:36-151 (column-major extract): one column per capture group.  Per row:
// find() the leftmost match, then run the program again from the match start with
// that group's id (regexec.inl:465-469: start window [begin, begin+1)); the row is
// null unless the recorded range is non-empty (extract.cu:144-145).
// Returns the number of groups; a pattern without groups yields no columns (extract.cu:95-100).
int orc_extract(const orc_col* c, const int32_t* prog, orc_col*** cols_out) {
  Prog p(prog);
  int groups = prog[2];
  *cols_out = nullptr;
  if (groups <= 0 || c->rows == 0) return 0;
  orc_col** cols = (orc_col**)malloc(sizeof(orc_col*) * groups);
  for (int g = 0; g < groups; ++g) {
    Builder b;
    for (int64_t r = 0; r < c->rows; ++r) {
      if (!c->is_valid(r)) {
        b.add_null();
        continue;
      }
      View v = make_view(*c, r);
      int begin = 0, end = (int)v.nchars;
      int res = re_find(p, v, begin, end);
      if (res > 0) {
        end = begin + 1;
        res = nfa_run(p, v, begin, end, g + 1);
      }
      if (res > 0 && begin >= 0 && end > begin) {
        unsigned s = byte_pos(v, (unsigned)begin), e = byte_pos(v, (unsigned)end);
        b.add(v.d + s, e - s);
      } else {
        b.add_null();
      }
    }
    cols[g] = b.finish();
  }
  *cols_out = cols;
  return groups;
}

// findall.cu:39-96,99-179 (column-major findall): column k holds every row's k-th match, found by
// the count_re walk (an empty match advances one character); rows with fewer matches and null
// rows are null, an empty match is an empty string; no match anywhere -> one all-null column.
int orc_findall(const orc_col* c, const int32_t* prog, orc_col*** cols_out) {
  Prog p(prog);
  *cols_out = nullptr;
  if (c->rows == 0) return 0;
  std::vector<std::vector<std::pair<int, int>>> spans(c->rows);  // char positions
  int ncols = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) continue;
    View v = make_view(*c, r);
    int nchars = (int)v.nchars, spos = 0, epos = nchars;
    while (spos <= nchars) {
      if (re_find(p, v, spos, epos) <= 0) break;
      spans[r].push_back({spos, epos});
      spos = epos > spos ? epos : spos + 1;
      epos = nchars;
    }
    ncols = std::max(ncols, (int)spans[r].size());
  }
  int nout = ncols ? ncols : 1;
  orc_col** cols = (orc_col**)malloc(sizeof(orc_col*) * nout);
  for (int k = 0; k < nout; ++k) {
    Builder b;
    for (int64_t r = 0; r < c->rows; ++r) {
      if (!c->is_valid(r) || k >= (int)spans[r].size()) {
        b.add_null();
        continue;
      }
      View v = make_view(*c, r);
      unsigned s = byte_pos(v, (unsigned)spans[r][k].first), e = byte_pos(v, (unsigned)spans[r][k].second);
      b.add(v.d + s, e > s ? e - s : 0);
    }
    cols[k] = b.finish();
  }
  *cols_out = cols;
  return nout;
}

// backref.h:31-57 parse_backrefs: every backslash followed by digits is a reference; returns the
// template without them and (index, byte position in the stripped template) per reference
static std::string parse_backrefs(const char* repl, std::vector<std::pair<int, int>>& refs) {
  std::string out;
  for (const char* p = repl; *p;) {
    if (*p == '\\' && p[1] >= '0' && p[1] <= '9') {
      const char* q = p + 1;
      while (*q >= '0' && *q <= '9') ++q;
      refs.push_back({atoi(p + 1), (int)out.size()});
      p = q;
    } else {
      out.push_back(*p++);
    }
  }
  return out;
}
// replace_backref.cu:36-125,128-207: every match is replaced by the template with the capture
// groups of THAT match filled in (each by a program run anchored at the match start, group id =
// reference number; 0 = the whole anchored match).  A match of length zero would repeat forever
// in the reference (begin = end, replace_backref.cu:112); the walk stops there (the product
// rejects such patterns).
orc_col* orc_replace_with_backrefs(const orc_col* c, const int32_t* prog, const char* repl) {
  Prog p(prog);
  std::vector<std::pair<int, int>> refs;
  std::string tmpl = parse_backrefs(repl, refs);
  Builder b;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) {
      b.add_null();
      continue;
    }
    View v = make_view(*c, r);
    int nchars = (int)v.nchars, begin = 0, end = nchars;
    unsigned lpos = 0;
    while (re_find(p, v, begin, end) > 0) {
      unsigned mb = byte_pos(v, (unsigned)begin);
      b.chars.insert(b.chars.end(), v.d + lpos, v.d + mb);
      int il = 0;
      for (auto& ref : refs) {
        b.chars.insert(b.chars.end(), tmpl.begin() + il, tmpl.begin() + ref.second);
        il = ref.second;
        int spos = begin, epos = begin + 1;
        if (nfa_run(p, v, spos, epos, ref.first) <= 0 || spos < 0 || epos <= spos) continue;
        unsigned x = byte_pos(v, (unsigned)spos), y = byte_pos(v, (unsigned)epos);
        b.chars.insert(b.chars.end(), v.d + x, v.d + y);
      }
      b.chars.insert(b.chars.end(), tmpl.begin() + il, tmpl.end());
      lpos = byte_pos(v, (unsigned)end);
      if (end == begin) break;
      begin = end;
      end = nchars;
    }
    b.chars.insert(b.chars.end(), v.d + lpos, v.d + v.bytes);
    b.close_row();
  }
  return b.finish();
}

// NVCategory.cu:220-304: sort (null first, bytewise, shorter-is-less), unique, rank
static int key_cmp(const orc_col* c, int64_t a, int64_t b) {  // custring.inl:240-261
  const uint8_t *pa = c->chars.data() + c->off[a], *pb = c->chars.data() + c->off[b];
  size_t la = (size_t)(c->off[a + 1] - c->off[a]), lb = (size_t)(c->off[b + 1] - c->off[b]);
  int m = memcmp(pa, pb, std::min(la, lb));
  if (m) return m;
  return la < lb ? -1 : (la > lb ? 1 : 0);
}
orc_col* orc_category(const orc_col* c, int32_t* values) {
  std::vector<int64_t> idx(c->rows);
  std::iota(idx.begin(), idx.end(), 0);
  auto less = [&](int64_t a, int64_t b) {
    bool va = c->is_valid(a), vb = c->is_valid(b);
    if (!va || !vb) return vb && !va;
    return key_cmp(c, a, b) < 0;
  };
  std::stable_sort(idx.begin(), idx.end(), less);
  Builder keys;
  int32_t rank = -1;
  for (int64_t i = 0; i < c->rows; ++i) {
    int64_t r = idx[i];
    bool fresh = i == 0;
    if (!fresh) {
      int64_t q = idx[i - 1];
      bool vr = c->is_valid(r), vq = c->is_valid(q);
      fresh = (vr != vq) || (vr && key_cmp(c, q, r) != 0);
    }
    if (fresh) {
      ++rank;
      if (c->is_valid(r))
        keys.add(c->chars.data() + c->off[r], (size_t)(c->off[r + 1] - c->off[r]));
      else
        keys.add_null();
    }
    values[r] = rank;
  }
  return keys.finish();
}

// tokens.cu:41-76,79-155. delimiter NULL -> char <= ' ', else any char of it
orc_col* orc_tokenize(const orc_col* c, const char* delim) {
  Builder b;
  View dv{(const uint8_t*)delim, delim ? (unsigned)strlen(delim) : 0, 0};
  if (delim) dv.nchars = count_chars(dv.d, dv.bytes);
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->is_valid(r)) continue;
    View v = make_view(*c, r);
    int nchars = (int)v.nchars, spos = 0, epos = nchars, pos = 0;
    bool spaces = true;
    for (;;) {
      // next_token
      if (spos >= nchars) break;
      for (; pos < nchars; ++pos) {
        Char ch = char_at(v, (unsigned)pos);
        bool isd = delim ? find_char(dv, ch) >= 0 : (ch <= ' ');
        if (spaces == isd) {
          if (spaces) spos = pos + 1;
          else epos = pos + 1;
          continue;
        }
        spaces = !spaces;
        if (spaces) {
          epos = pos;
          break;
        }
      }
      if (!(spos < epos)) break;
      unsigned s = byte_pos(v, (unsigned)spos), e = byte_pos(v, (unsigned)epos);
      b.add(v.d + s, e - s);
      spos = epos + 1;
      epos = nchars;
      ++pos;
    }
  }
  return b.finish();
}

// combine.cu join(separator, narep="") as used by ngram.cu:51-52
static orc_col* join_all(const orc_col* c, const char* sep) {
  Builder b;
  size_t sb = strlen(sep);
  for (int64_t r = 0; r < c->rows; ++r) {
    if (c->is_valid(r))
      b.chars.insert(b.chars.end(), c->chars.data() + c->off[r], c->chars.data() + c->off[r + 1]);
    if (r + 1 < c->rows) b.chars.insert(b.chars.end(), sep, sep + sb);
  }
  b.close_row();
  return b.finish();
}
static orc_col* copy_col(const orc_col* c) { return new orc_col(*c); }
// ngram.cu:32-110
orc_col* orc_ngrams(const orc_col* c, unsigned n, const char* sep) {
  if (n == 0) n = 2;
  if (!sep) sep = "";
  if (c->rows == 0) return copy_col(c);
  std::vector<int64_t> keep;
  for (int64_t r = 0; r < c->rows; ++r)
    if (c->is_valid(r) && c->off[r + 1] > c->off[r]) keep.push_back(r);
  if (keep.size() <= n) return join_all(c, sep);
  if (n == 1) return copy_col(c);
  size_t sb = strlen(sep);
  Builder b;
  for (size_t i = 0; i + n <= keep.size(); ++i) {
    for (unsigned k = 0; k < n; ++k) {
      int64_t r = keep[i + k];
      b.chars.insert(b.chars.end(), c->chars.data() + c->off[r], c->chars.data() + c->off[r + 1]);
      if (k + 1 < n) b.chars.insert(b.chars.end(), sep, sep + sb);
    }
    b.close_row();
  }
  return b.finish();
}

// helpers for tests: unicode table access
const uint8_t* orc_flags_table() { return orc_unicode_flags; }
const uint16_t* orc_cases_table() { return orc_charcases; }

}  // extern "C"

// ---- synthetic workloads (spec in include/cs_synth_spec.h) ----
#include "cs_synth_spec.h"
extern "C" orc_col* orc_synth(int kind, int64_t first_row, int64_t rows, uint64_t seed, int64_t param) {
  Builder b;
  std::vector<uint8_t> tmp(256);
  for (int64_t i = 0; i < rows; ++i) {
    int64_t r = first_row + i;
    if (cs_synth_is_null(kind, seed, r)) {
      b.add_null();
      continue;
    }
    int n = cs_synth_row(kind, seed, r, param, tmp.data());
    b.add(tmp.data(), (size_t)n);
  }
  return b.finish();
}
extern "C" uint64_t orc_digest(const orc_col* c) {
  uint64_t d = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    bool ok = c->is_valid(r);
    d += cs_digest_row((uint64_t)r, ok ? c->chars.data() + c->off[r] : nullptr,
                       ok ? (int)(c->off[r + 1] - c->off[r]) : 0, ok);
  }
  return d;
}

#include "oracle_round2.inc"
