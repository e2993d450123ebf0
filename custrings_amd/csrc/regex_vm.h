// Regex executor: ordered-thread (Pike) simulation of a csrx::Program over one
// row, in byte-offset space.  `__host__ __device__`: runs in the HIP kernels
// (program image + thread lists staged in LDS) and in tests/rowemu on the CPU.
//
// Semantics restated from the reference's dreprog::regexec
// (/root/reference/cpp/src/regex/regexec.inl:204-442):
//  * threads are an ORDERED list (priority = list order); an instruction is
//    activated at most once per step, first activation wins (:26-108);
//  * new start threads are appended, lowest priority, at every position inside
//    the start window while no match has been recorded (:260-267);
//  * on END the match (thread start, current position) is recorded, all
//    lower-priority threads are dropped, higher-priority survivors may
//    overwrite it later (:423-427);
//  * stops at the end of the row or at an embedded NUL character (:440).
// The reference expands the non-consuming instructions by repeated list passes
// (:274-362); here the same ordered closure is computed by an explicit-stack
// depth-first walk (identical order: an OR visits its `right` branch first),
// done once per step when a thread advances, using one character of
// look-ahead for the position-dependent ^ $ \b \B tests.  Thread starts are
// kept as byte offsets, so no char<->byte conversions are needed afterwards.
#pragma once
#include <stdint.h>

#include "row_ops.h"

namespace csvm {

using csrow::Char;

enum {
  I_CHAR = 0177, I_RBRA = 0201, I_LBRA = 0202, I_OR = 0204, I_ANY = 0300, I_ANYNL = 0301,
  I_BOL = 0303, I_EOL = 0304, I_CCLASS = 0305, I_NCCLASS = 0306, I_BOW = 0307, I_NBOW = 0310,
  I_END = 0377
};

// Device image = program blob (regex_program.h) + an "extras" section at word
// offset blob[7]:  [0] prefilter usable  [1..4] first-char ASCII bitmap
//                  [5] first char may be non-ASCII   [6..9] \w ASCII bitmap
//                  [10 + 4*cls ..] per-class ASCII membership bitmap
struct ProgView {
  const int32_t* insts;
  const int32_t* starts;
  const int32_t* cls_off;
  const int32_t* cls_data;
  const uint32_t* extra;
  const uint8_t* flags;  // unicode flag table (64 KiB, global memory)
  int ninst;
};
CS_HD ProgView make_view(const int32_t* image, const uint8_t* flags) {
  ProgView v;
  v.ninst = image[3];
  v.insts = image + 8;
  v.starts = v.insts + 4 * v.ninst;
  v.cls_off = v.starts + image[4];
  v.cls_data = v.cls_off + image[5] + 1;
  v.extra = (const uint32_t*)(image + image[7]);
  v.flags = flags;
  return v;
}

CS_HD bool bm_test(const uint32_t* bm, unsigned c) { return (bm[c >> 5] >> (c & 31)) & 1u; }

CS_HD bool is_word(const ProgView& P, Char c) {
  if (c < 128) return bm_test(P.extra + 6, c);
  unsigned u = csrow::packed_to_cp(c);
  return u < 0x10000 && (P.flags[u] & 15);
}
// dreclass::is_match, regexec.inl:127-155
CS_HD bool class_match(const ProgView& P, int cls, Char ch) {
  if (ch < 128) return bm_test(P.extra + 10 + 4 * cls, ch);
  const int32_t* c = P.cls_data + P.cls_off[cls];
  int nr = P.cls_off[cls + 1] - P.cls_off[cls] - 1;
  int builtins = c[0];
  for (int i = 0; i < nr; i += 2)
    if (ch >= (Char)c[1 + i] && ch <= (Char)c[2 + i]) return true;
  if (!builtins) return false;
  unsigned u = csrow::packed_to_cp(ch);
  if (u > 0xFFFF) return false;
  unsigned f = P.flags[u];
  bool alnum = (f & 15) != 0;
  if ((builtins & 1) && (ch == '_' || alnum)) return true;
  if ((builtins & 2) && (f & 16)) return true;
  if ((builtins & 4) && (f & 4)) return true;
  if ((builtins & 8) && (ch != '\n' && ch != '_' && !alnum)) return true;
  if ((builtins & 16) && !(f & 16)) return true;
  if ((builtins & 32) && (ch != '\n' && !(f & 4))) return true;
  return false;
}

// Per-thread scratch: 6*ninst (+ mask words when ninst > 64) 32-bit slots,
// addressed as mem[slot * stride] so that the lanes of a wave interleave
// (bank-conflict-free in LDS, coalesced in global scratch).
CS_HD int vm_slots(int ninst) { return 6 * ninst + (ninst > 64 ? (ninst + 31) / 32 : 0); }

template <bool SMALL>  // SMALL: ninst <= 64, visited set lives in a register
struct Vm {
  const ProgView& P;
  uint32_t* mem;
  int stride;
  const uint8_t* s;
  int n;
  int N;  // ninst

  uint64_t seen;  // SMALL
  int lst;        // list being built: 0/1
  int cnt;        // entries in the list being built

  CS_HD Vm(const ProgView& p, uint32_t* m, int st, const uint8_t* row, int bytes)
      : P(p), mem(m), stride(st), s(row), n(bytes), N(p.ninst), seen(0), lst(0), cnt(0) {}

  CS_HD uint32_t& id_at(int l, int k) { return mem[(l * N + k) * stride]; }
  CS_HD uint32_t& sx_at(int l, int k) { return mem[(2 * N + l * N + k) * stride]; }
  CS_HD uint32_t& stk(int k) { return mem[(4 * N + k) * stride]; }
  CS_HD uint32_t& mask_word(int k) { return mem[(6 * N + k) * stride]; }

  CS_HD void begin_list(int l) {
    lst = l;
    cnt = 0;
    if (SMALL) {
      seen = 0;
    } else {
      for (int k = 0; k < (N + 31) / 32; ++k) mask_word(k) = 0;
    }
  }
  CS_HD bool test_and_set(int id) {
    if (SMALL) {
      uint64_t b = 1ull << id;
      bool was = (seen & b) != 0;
      seen |= b;
      return was;
    } else {
      uint32_t& w = mask_word(id >> 5);
      uint32_t b = 1u << (id & 31);
      bool was = (w & b) != 0;
      w |= b;
      return was;
    }
  }

  // ordered epsilon-closure of `inst` into the list being built; position
  // tests use: at = byte offset, pc = char before, cc = char at `at`
  CS_HD void closure(int inst, uint32_t start, int at, Char pc, Char cc) {
    int sp = 0;
    stk(sp++) = (uint32_t)inst;
    while (sp > 0) {
      int id = (int)stk(--sp);
      if (test_and_set(id)) continue;
      const int32_t* in = P.insts + 4 * id;
      int type = in[0];
      switch (type) {
        case I_OR:
          stk(sp++) = (uint32_t)in[2];  // left: lower priority, visited second
          stk(sp++) = (uint32_t)in[1];  // right: preferred
          break;
        case I_LBRA:
        case I_RBRA:
          stk(sp++) = (uint32_t)in[2];
          break;
        case I_BOL:
          if (at == 0 || ((Char)in[1] == '^' && pc == '\n')) stk(sp++) = (uint32_t)in[2];
          break;
        case I_EOL:
          if (cc == 0 || ((Char)in[1] == '$' && cc == '\n')) stk(sp++) = (uint32_t)in[2];
          break;
        case I_BOW:
        case I_NBOW:
          if ((is_word(P, cc) != is_word(P, pc)) == (type == I_BOW)) stk(sp++) = (uint32_t)in[2];
          break;
        default:  // consuming instructions, END, and unknown opcodes wait in the list
          id_at(lst, cnt) = (uint32_t)id;
          sx_at(lst, cnt) = start;
          ++cnt;
          break;
      }
    }
  }

  CS_HD Char char_at(int i, unsigned& w) const {
    if (i >= n) {
      w = 1;
      return 0;
    }
    Char c;
    w = csrow::decode_at(s, i, n, c);
    if (w == 0) w = 1;
    // a multi-byte sequence cut off by the end of the row (malformed input) must not carry the
    // scan position past the row: match spans stay inside [0, n], so the size pass and the
    // write pass of every replace kernel agree byte for byte
    if (i + (int)w > n) w = (unsigned)(n - i);
    return c;
  }
  CS_HD Char char_before(int i) const {
    if (i <= 0) return 0;
    int q = i - 1;
    while (q > 0 && csrow::is_cont(s[q])) --q;
    Char c;
    csrow::decode_at(s, q, n, c);
    return c;
  }

  // Leftmost match whose start lies in [from, win_end).  Returns 1 and the
  // match span [mb, me) in byte offsets, or 0.
  CS_HD int find(int from, int win_end, int& mb, int& me) {
    int match = 0;
    int pos = from;
    unsigned w = 1, wn = 1;
    Char pc = char_before(pos);
    Char c = char_at(pos, w);
    Char cn = c ? char_at(pos + (int)w, wn) : 0;
    const bool prefilter = (P.extra[0] & 1u) != 0;
    const bool bol_jump = (P.extra[0] & 4u) != 0;  // (the first instruction is a multi-line `^`)
    int cur = 0;
    begin_list(cur);  // threads that advanced into `pos` (none yet)
    for (;;) {
      if (match == 0 && pos < win_end) {
        if (cnt == 0 && bol_jump && pos != 0) {
          // the reference's search for the next start (regexec.inl:233-246): the byte behind the next line feed at or after the
          // previous byte -- found by length, NUL bytes or not
          int q = pos - 1;
          while (q < n && s[q] != '\n') ++q;
          if (q >= n) break;
          if (q + 1 != pos) {
            pos = q + 1;
            pc = char_before(pos);
            c = char_at(pos, w);
            cn = c ? char_at(pos + (int)w, wn) : 0;
            begin_list(cur);
          }
        }
        if (cnt == 0 && prefilter) {
          // no live thread: a start thread can only survive on a character that
          // can begin a match, so skip ahead to the next such character (this
          // generalises the reference's first-CHAR shortcut, regexec.inl:220-258)
          int q = pos;
          while (q < win_end && q < n) {
            uint8_t b = s[q];
            if (b == 0 && !(P.extra[0] & 2u)) break;  // (bit 1: the first instruction is a literal -- the reference's jump to it passes NUL bytes)
            if (b < 128 ? bm_test(P.extra + 1, b) : (P.extra[5] != 0 && !csrow::is_cont(b))) break;
            ++q;
          }
          if (q != pos) {
            pos = q;
            pc = char_before(pos);
            c = char_at(pos, w);
            cn = c ? char_at(pos + (int)w, wn) : 0;
            begin_list(cur);
          }
        }
        if (pos < win_end)
          for (int i = 0; P.starts[i] >= 0; ++i) closure(P.starts[i], (uint32_t)pos, pos, pc, c);
      }
      const int ncur = cnt;
      if (ncur == 0 && (match || pos >= win_end)) break;
      // consume c; threads that advance are closed over at the next position
      const int nxt = cur ^ 1;
      const int npos = pos + (int)w;
      begin_list(nxt);
      for (int k = 0; k < ncur; ++k) {
        int id = (int)id_at(cur, k);
        uint32_t st = sx_at(cur, k);
        const int32_t* in = P.insts + 4 * id;
        bool go = false;
        int type = in[0];
        if (type == I_CHAR) go = (Char)in[1] == c;
        else if (type == I_ANY) go = c != '\n';
        else if (type == I_ANYNL) go = true;
        else if (type == I_CCLASS) go = class_match(P, in[1], c);
        else if (type == I_NCCLASS) go = !class_match(P, in[1], c);
        else if (type == I_END) {
          match = 1;
          mb = (int)st;
          me = pos;
          break;
        }
        if (go) closure(in[2], st, npos, c, cn);
      }
      if (c == 0) break;
      pos = npos;
      pc = c;
      c = cn;
      w = wn;
      cn = c ? char_at(pos + (int)w, wn) : 0;
      cur = nxt;
    }
    return match;
  }
};

// ---- capture groups (regexec.inl:204-442 with groupId != 0) -------------------------------
// The same ordered simulation with a (begin, end) range per thread instead of a start offset:
// a thread is born with (-1, -1), LBRA / RBRA of the wanted group set begin / end to the
// position they are expanded at (regexec.inl:296-307), END reports the range of the first
// thread in list order (regexec.inl:425-428).  Only used anchored at a known match start
// (dreprog::extract, regexec.inl:465-469: start window [begin, begin + 1)).
// Scratch: 12*ninst (+ mask words) 32-bit slots per thread, interleaved like Vm's.
CS_HD int gvm_slots(int ninst) { return 12 * ninst + (ninst > 64 ? (ninst + 31) / 32 : 0); }

template <bool SMALL>
struct GroupVm {
  const ProgView& P;
  uint32_t* mem;
  int stride;
  const uint8_t* s;
  int n;
  int N;
  uint64_t seen;
  int lst, cnt;
  Vm<true> text;  // character decoding only (no scratch touched)

  // Optional fast region (LDS on the device): the first kFastList entries of either list and the
  // first kFastStack stack entries live there, the rest in `mem`.  An anchored run keeps a handful
  // of threads alive, so the global arena is touched only by unusual programs.
  static constexpr int kFastList = 4, kFastStack = 6;
  static constexpr int kFastSlots = 3 * (2 * kFastList + kFastStack);
  uint32_t* fast = nullptr;
  int fstride = 0;

  CS_HD GroupVm(const ProgView& p, uint32_t* m, int st, const uint8_t* row, int bytes, uint32_t* fast_mem = nullptr, int fast_stride = 0)
      : P(p), mem(m), stride(st), s(row), n(bytes), N(p.ninst), seen(0), lst(0), cnt(0), text(p, nullptr, 0, row, bytes),
        fast(fast_mem), fstride(fast_stride) {}

  // lists: id | begin | end, two lists each; closure stack: id | begin | end, 2*ninst deep
  CS_HD uint32_t& list_at(int f, int l, int k) {
    if (fast && k < kFastList) return fast[((f * 2 + l) * kFastList + k) * fstride];
    return mem[(2 * f * N + l * N + k) * stride];
  }
  CS_HD uint32_t& id_at(int l, int k) { return list_at(0, l, k); }
  CS_HD uint32_t& bx_at(int l, int k) { return list_at(1, l, k); }
  CS_HD uint32_t& by_at(int l, int k) { return list_at(2, l, k); }
  CS_HD uint32_t& stk(int f, int k) {
    if (fast && k < kFastStack) return fast[(6 * kFastList + f * kFastStack + k) * fstride];
    return mem[(6 * N + f * 2 * N + k) * stride];
  }
  CS_HD uint32_t& mask_word(int k) { return mem[(12 * N + k) * stride]; }

  CS_HD void begin_list(int l) {
    lst = l;
    cnt = 0;
    if (SMALL) {
      seen = 0;
    } else {
      for (int k = 0; k < (N + 31) / 32; ++k) mask_word(k) = 0;
    }
  }
  CS_HD bool test_and_set(int id) {
    if (SMALL) {
      uint64_t b = 1ull << id;
      bool was = (seen & b) != 0;
      seen |= b;
      return was;
    } else {
      uint32_t& w = mask_word(id >> 5);
      uint32_t b = 1u << (id & 31);
      bool was = (w & b) != 0;
      w |= b;
      return was;
    }
  }
  CS_HD void closure(int inst, int group, int bx, int by, int at, Char pc, Char cc) {
    int sp = 0;
    auto push = [&](int id, int x, int y) {
      stk(0, sp) = (uint32_t)id;
      stk(1, sp) = (uint32_t)x;
      stk(2, sp) = (uint32_t)y;
      ++sp;
    };
    push(inst, bx, by);
    while (sp > 0) {
      --sp;
      const int id = (int)stk(0, sp), x = (int)stk(1, sp), y = (int)stk(2, sp);
      if (test_and_set(id)) continue;
      const int32_t* in = P.insts + 4 * id;
      const int type = in[0];
      switch (type) {
        case I_OR:
          push(in[2], x, y);  // left: lower priority, visited second
          push(in[1], x, y);  // right: preferred
          break;
        case I_LBRA:
          push(in[2], in[1] == group ? at : x, y);
          break;
        case I_RBRA:
          push(in[2], x, in[1] == group ? at : y);
          break;
        case I_BOL:
          if (at == 0 || ((Char)in[1] == '^' && pc == '\n')) push(in[2], x, y);
          break;
        case I_EOL:
          if (cc == 0 || ((Char)in[1] == '$' && cc == '\n')) push(in[2], x, y);
          break;
        case I_BOW:
        case I_NBOW:
          if ((is_word(P, cc) != is_word(P, pc)) == (type == I_BOW)) push(in[2], x, y);
          break;
        default:
          id_at(lst, cnt) = (uint32_t)id;
          bx_at(lst, cnt) = (uint32_t)x;
          by_at(lst, cnt) = (uint32_t)y;
          ++cnt;
          break;
      }
    }
  }
  // The program run from byte offset `from` only (threads are born there and nowhere else);
  // on a match (gb, ge) is the range recorded for `group` (>= 1), byte offsets or -1.
  CS_HD int run(int from, int group, int& gb, int& ge) {
    int match = 0;
    int pos = from;
    unsigned w = 1, wn = 1;
    Char pc = text.char_before(pos);
    Char c = text.char_at(pos, w);
    Char cn = c ? text.char_at(pos + (int)w, wn) : 0;
    int cur = 0;
    begin_list(cur);
    for (int i = 0; P.starts[i] >= 0; ++i) closure(P.starts[i], group, -1, -1, pos, pc, c);
    for (;;) {
      const int ncur = cnt;
      if (ncur == 0) break;
      const int nxt = cur ^ 1;
      const int npos = pos + (int)w;
      begin_list(nxt);
      for (int k = 0; k < ncur; ++k) {
        const int id = (int)id_at(cur, k);
        const int x = (int)bx_at(cur, k), y = (int)by_at(cur, k);
        const int32_t* in = P.insts + 4 * id;
        bool go = false;
        const int type = in[0];
        if (type == I_CHAR) go = (Char)in[1] == c;
        else if (type == I_ANY) go = c != '\n';
        else if (type == I_ANYNL) go = true;
        else if (type == I_CCLASS) go = class_match(P, in[1], c);
        else if (type == I_NCCLASS) go = !class_match(P, in[1], c);
        else if (type == I_END) {
          match = 1;
          gb = x;
          ge = y;
          break;
        }
        if (go) closure(in[2], group, x, y, npos, c, cn);
      }
      if (c == 0) break;
      pos = npos;
      pc = c;
      c = cn;
      w = wn;
      cn = c ? text.char_at(pos + (int)w, wn) : 0;
      cur = nxt;
    }
    return match;
  }
};

// ---- row-level drivers (count.cu:36-56,168-196 ; replace.cu:39-107) -----------
template <class VM>
CS_HD int row_contains_re(VM& vm, bool anchored) {
  int mb, me;
  // match(): start window is [0,1) even for an empty row (count.cu:51)
  return vm.find(0, anchored ? 1 : vm.n, mb, me);
}
template <class VM>
CS_HD int row_count_re(VM& vm) {
  int k = 0, from = 0;
  while (from <= vm.n) {
    int mb, me;
    if (!vm.find(from, vm.n, mb, me)) break;
    ++k;
    if (me > mb) {
      from = me;
    } else {  // empty match: step one character
      unsigned w;
      vm.char_at(mb, w);
      from = mb + (int)w;
    }
  }
  return k;
}
// replace_backref.cu:36-125: walks the matches of a row and sends the output through out(ptr, len):
// the text before each match, then the template with the capture groups of that match filled in.
// Matches are never empty where it matters (the host rejects patterns that can match the empty string
// when the DFA knows: the reference would not terminate on them).
struct BackrefTemplate {
  static constexpr int kMaxRefs = 16;
  const uint8_t* text;  // template without the references
  int bytes;
  int nrefs;
  int idx[kMaxRefs];  // reference number
  int pos[kMaxRefs];  // byte position in `text`
  int groups;         // capture groups of the program
};
// walk(f): calls f(mb, me) for every match of the row in replace order (non-empty matches: the next search
// starts at me); group(mb, g, x, y): the range of group g (0 = whole match) in the program run anchored at mb.
template <class Walk, class Group, class Out>
CS_HD void row_backrefs(const uint8_t* p, int n, const BackrefTemplate& t, Walk&& walk, Group&& group, Out&& out) {
  int lpos = 0;
  walk([&](int mb, int me) {
    out(p + lpos, mb - lpos);
    int il = 0;
    for (int j = 0; j < t.nrefs; ++j) {
      out(t.text + il, t.pos[j] - il);
      il = t.pos[j];
      int x = -1, y = -1;
      if (t.idx[j] <= t.groups && group(mb, t.idx[j], x, y) && x >= 0 && y > x) out(p + x, y - x);
    }
    out(t.text + il, t.bytes - il);
    lpos = me;
  });
  out(p + lpos, n - lpos);
}
// the match walk on successive find() calls (list simulators); stops at an empty match
template <class Find, class F>
CS_HD void walk_matches_by_find(Find&& find, F&& f, int from = 0) {
  for (;;) {
    int mb = 0, me = 0;
    if (!find(from, mb, me)) break;
    f(mb, me);
    if (me <= mb) break;
    from = me;
  }
}
// findall.cu:61-77: the count_re walk reporting every match span; emit(k, mb, me) returns false to stop.
template <class VM, class Emit>
CS_HD int row_findall(VM& vm, Emit&& emit) {
  int k = 0, from = 0;
  while (from <= vm.n) {
    int mb, me;
    if (!vm.find(from, vm.n, mb, me)) break;
    if (!emit(k, mb, me)) return k + 1;
    ++k;
    if (me > mb) {
      from = me;
    } else {  // empty match: step one character
      unsigned w;
      vm.char_at(mb, w);
      from = mb + (int)w;
    }
  }
  return k;
}
// Walks the successive matches exactly as replace_re does; emit(mb, me) per
// replacement, `reps` identical zero-length replacements are reported at once.
template <class VM, class Emit>
CS_HD void row_replace_matches(VM& vm, int maxrepl, Emit&& emit) {
  // maxrepl < 0 means "up to nchars replacements" (replace.cu:66-67).  Every
  // non-empty match consumes at least one character, so that budget can only
  // bind on a zero-length match (after nchars advancing matches the row is used
  // up and find() fails by itself); the character count is therefore taken lazily.
  int done = 0;
  int from = 0;
  for (;;) {
    if (maxrepl >= 0 && done >= maxrepl) break;
    int mb, me;
    if (!vm.find(from, vm.n, mb, me)) break;
    if (me == mb && mb == from) {
      // a zero-length match does not advance the search (replace.cu:91-93):
      // the same match repeats until the budget is spent
      int left = (maxrepl < 0 ? csrow::count_chars(vm.s, vm.n) : maxrepl) - done;
      if (left > 0) emit(mb, me, left);
      return;
    }
    emit(mb, me, 1);
    from = me;
    ++done;
  }
}

// extract.cu:36-66,139-146: the byte span [x, y) of capture group `group` (1-based) in the match
// that starts at byte offset `mb` (found by find()); false = the row's result is null (the group
// took no part in the match, or matched the empty string: extract.cu:144-145).
template <class GVM>
CS_HD bool row_group_span(GVM& g, int mb, int group, int& x, int& y) {
  int gb = -1, ge = -1;
  if (!g.run(mb, group, gb, ge)) return false;
  if (gb < 0 || ge <= gb) return false;
  x = gb;
  y = ge;
  return true;
}

}  // namespace csvm
