// Bit-parallel form of a compiled regex program ("bit program"): the matches of a plain-ASCII row by integer arithmetic on
// per-class bit masks of the row -- no automaton, no table walk per byte.  For patterns whose candidate bytes are
// everywhere (alternations of word-bounded literals, small character classes in a `+` loop: the reference's own gtest
// pattern (\bin\b)|(\ba\b)|(\bthe\b), cpp/tests/test_replace.cpp:40), where the tagged DFA has to walk every byte of
// every row in a dependent chain of table reads.
//
// What converts (regex_bits.cpp, from the Reprog-identical instruction stream): a program without loops whose paths from
// the start to END number at most kMaxAlts -- alternations and optional parts expand into ALTERNATIVES, in the order in
// which the reference's executor prefers them (regexec.inl:204-442: the OR's preferred branch first) --, every path a
// sequence of single-character items (literal, class, `.`) and assertions (\b \B ^ $ \A \Z) of fixed length 1..31; or ONE
// such path that ends in a greedy `+` loop over its last item (`[aeiou]+`, `#\w+`), with assertions behind the loop at most
// (`[^ ]+$`, `\w+\b`: the TAIL).  At most kMaxClasses distinct ASCII member sets over all items.
//
// Semantics (what the reference computes, restated on masks).  Row of n bytes, all ASCII, none NUL; bit i of a class mask
// = byte i is a member; cursor q in 0..n lies in front of byte q.  An alternative with items (X_i at offset o_i) matches
// at start p iff every X_i holds at p + o_i: A_j = AND_i (X_i >> o_i), all starts at once.  The executor prefers the
// leftmost start and, at one start, the first alternative in priority order that matches (threads of an earlier start,
// then of an earlier branch, come first in its list and cut the later ones off at END): sel_j = A_j & ~(A_1 | ... |
// A_{j-1}); the match at p has alternative j's length.  Successive matches do not overlap: the next search starts at
// the end of the last match (replace.cu:91-93, count.cu:168-250) -- a loop over the row's matches, lowest start first.
// With a trailing `+` the match runs to the end of the run of its last class (greedy, nothing follows that could ask
// for less).  With a tail the loop's exits are tried longest first (the executor prefers the thread that stays in the loop):
// the match ends at the LAST cursor of the run at which every assertion of the tail holds, and a start whose run has no such
// cursor matches nothing (`[^ ]+$`: only the run that reaches the row's end -- the automaton's scan tried every start of every
// run to its end: 35 ms on the C3 column).  An alternative cannot match the empty string (such programs do not convert).
//
// Image (int32 words):
//   [0] magic 'CSBP' [1] classes K [2] flags [3] alternatives J [4] words in all [5] class of the trailing `+` (-1: none)
//   [6] class holding the word characters (for \b \B; -1: not needed) [7] class holding '\n' (multi-line ^ $; -1)
//   [8..39]  128 bytes: for every ASCII byte the set of classes it belongs to (bit k = class k)
//   [40..]   per alternative: items | length << 8, then one word per item: kind | class << 8 | offset << 16
//   (F_TAIL) the image's last word: assertions behind the `+` loop, count | kind << 8 | kind << 16 | kind << 24
#pragma once
#include <stdint.h>

#include "regex_tdfa.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define CSBITS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)  // (the program words are the same for every lane: scalar control flow)
#else
#define CSBITS_UNIFORM(x) (x)
#endif

namespace csbits {

using cstd::U128;

constexpr int32_t kMagic = 0x50425343;  // "CSBP"
constexpr int kMaxClasses = 8, kMaxAlts = 16, kMaxItems = 32, kMaxLen = 31, kHeaderWords = 8, kTableWords = 32;
constexpr int kMaxWords = 40 + kMaxAlts * (1 + kMaxItems);
constexpr int kMaxRowBytes = 95;  // cursors 0..n fit the 96-bit masks
enum { F_WORD = 1, F_BOL = 2, F_EOL = 4, F_BOL_MULTI = 8, F_EOL_MULTI = 16, F_PURE_PLUS = 32, F_PLUS = 64, F_SAME_LEN = 128,
       // the program is ONE class, once or in a `+` loop, whose membership is decided byte by byte on ANY text: a literal ASCII
       // byte, a set of ASCII bytes and ranges (no builtin classes: \w \s \d reach into the non-ASCII characters), its negation
       // or `.` -- F_HIGH_MEMBER: every non-ASCII character, i.e. every byte >= 0x80, is a member (cs_runs.hip)
       F_BYTE_CLASS = 256, F_HIGH_MEMBER = 512,
       // ... or ONE class of ASCII ranges and builtin classes (`\w+`, `[\w.]`, `\S+`, `[^\d]`): byte by byte on text WITHOUT bytes >= 0x80;
       // a non-ASCII character is a member through the unicode flags alone -- bits 16..21 the class's builtins (regex_vm.h:
       // class_match), bit 22 the class is negated: cs_runs.hip takes tiles with such bytes row by row, decoding as the reference does
       F_FLAG_CLASS = 1024,
       F_TAIL = 2048,        // assertions behind the trailing `+` loop (the image's last word)
       F_PURE_TAIL = 4096,  // ... behind `[set]+` alone (`[^ ]+$`): the matches from the tail's cursors, a step per MATCH
       // the program's first instruction is a literal character: the reference's search for the next start jumps to it by length
       // (regexec.inl:220-232), over NUL bytes -- a NUL ends nothing for `a` or `a+` (cs_runs.hip)
       F_CHAR_FIRST = 8192 };
enum { K_CLASS = 0, K_BOW = 1, K_NBOW = 2, K_BOL = 3, K_EOL = 4, K_BOL_MULTI = 5, K_EOL_MULTI = 6 };

struct View {
  const int32_t* img;
  int K, J, plus_cls, word_cls, nl_cls;
  uint32_t flags;
};
CS_HD View make_view(const int32_t* img) {
  View v;
  v.img = img;
  v.K = img[1];
  v.flags = (uint32_t)img[2];
  v.J = img[3];
  v.plus_cls = img[5];
  v.word_cls = img[6];
  v.nl_cls = img[7];
  return v;
}
CS_HD bool offered(const int32_t* img, int words) { return words >= 41 && img[0] == kMagic; }

CS_HD U128 shr(U128 a, int k) {  // k in 0..63
  if (k == 0) return a;
  return cstd::u128((a.lo >> k) | (a.hi << (64 - k)), a.hi >> k);
}
CS_HD U128 shr1(U128 a) { return cstd::u128((a.lo >> 1) | (a.hi << 63), a.hi >> 1); }
CS_HD U128 bit_at(int q) { return q < 64 ? cstd::u128(1ull << q, 0) : cstd::u128(0, 1ull << (q - 64)); }
CS_HD unsigned test(U128 a, int q) { return (unsigned)((q < 64 ? a.lo >> q : a.hi >> (q - 64)) & 1ull); }

// Classification by table lookup.  The SPREAD table holds, for every ASCII byte, its class set with class k at bit 4k: the
// entries of four consecutive bytes, shifted by 0..3 and OR-ed, leave one nibble per class -- the four bytes' membership
// bits.  Sixteen bytes (four words, first byte lowest) give, for every class, sixteen bits: pair[m] = class 2m | class
// 2m+1 << 16.  (The low seven bits of a byte index the table: a piece with bytes >= 0x80 never uses the result.)
CS_HD uint32_t spread_entry(uint32_t set) {
  uint32_t e = 0;
  for (int k = 0; k < 8; ++k) e |= ((set >> k) & 1u) << (4 * k);
  return e;
}
CS_HD void classify16(const uint32_t* spread, uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t pair[4]) {
  const uint32_t q[4] = {x, y, z, w};
  uint32_t a[4];
  for (int j = 0; j < 4; ++j)
    a[j] = spread[q[j] & 127u] | (spread[(q[j] >> 8) & 127u] << 1) | (spread[(q[j] >> 16) & 127u] << 2) | (spread[(q[j] >> 24) & 127u] << 3);
  // byte m of lo_e / hi_e: bytes 0..7 / 8..15 of the piece for class 2m (lo_o / hi_o: class 2m + 1)
  const uint32_t lo_e = (a[0] & 0x0F0F0F0Fu) | ((a[1] & 0x0F0F0F0Fu) << 4), hi_e = (a[2] & 0x0F0F0F0Fu) | ((a[3] & 0x0F0F0F0Fu) << 4);
  const uint32_t lo_o = ((a[0] >> 4) & 0x0F0F0F0Fu) | (a[1] & 0xF0F0F0F0u), hi_o = ((a[2] >> 4) & 0x0F0F0F0Fu) | (a[3] & 0xF0F0F0F0u);
#if defined(__HIP_DEVICE_COMPILE__)
  // one byte permute per pair of classes: [lo_e.m, hi_e.m, lo_o.m, hi_o.m] out of the eight bytes of (hi, lo)
  const uint32_t e01 = __builtin_amdgcn_perm(hi_e, lo_e, 0x05010400u), e23 = __builtin_amdgcn_perm(hi_e, lo_e, 0x07030602u);
  const uint32_t o01 = __builtin_amdgcn_perm(hi_o, lo_o, 0x05010400u), o23 = __builtin_amdgcn_perm(hi_o, lo_o, 0x07030602u);
  // e01 = class 0 (16 bits) | class 2 << 16; o01 = class 1 | class 3 << 16; e23 = class 4 | class 6 << 16; o23 = class 5 | class 7 << 16
  pair[0] = __builtin_amdgcn_perm(o01, e01, 0x05040100u);  // class 0 | class 1 << 16
  pair[1] = __builtin_amdgcn_perm(o01, e01, 0x07060302u);  // class 2 | class 3 << 16
  pair[2] = __builtin_amdgcn_perm(o23, e23, 0x05040100u);
  pair[3] = __builtin_amdgcn_perm(o23, e23, 0x07060302u);
#else
  for (int m = 0; m < 4; ++m) {
    const uint32_t even = ((lo_e >> (8 * m)) & 255u) | (((hi_e >> (8 * m)) & 255u) << 8);
    const uint32_t odd = ((lo_o >> (8 * m)) & 255u) | (((hi_o >> (8 * m)) & 255u) << 8);
    pair[m] = even | (odd << 16);
  }
#endif
}

// A row's mask: 96 bits in three words (bit i of w[i / 32]: byte / cursor i).  The evaluation below runs on these -- the
// 2 x 64-bit form the kernels keep their match bits in costs four register operations where three do, and its shifts are
// 64-bit ones.
struct M96 {
  uint32_t a, b, c;
};
CS_HD M96 m96(uint32_t a, uint32_t b, uint32_t c) {
  M96 r;
  r.a = a;
  r.b = b;
  r.c = c;
  return r;
}
CS_HD M96 m_and(M96 x, M96 y) { return m96(x.a & y.a, x.b & y.b, x.c & y.c); }
CS_HD M96 m_or(M96 x, M96 y) { return m96(x.a | y.a, x.b | y.b, x.c | y.c); }
CS_HD M96 m_andn(M96 x, M96 y) { return m96(x.a & ~y.a, x.b & ~y.b, x.c & ~y.c); }  // x & ~y
CS_HD bool m_any(M96 x) { return (x.a | x.b | x.c) != 0; }
CS_HD uint32_t funnel_r(uint32_t hi, uint32_t lo, unsigned k) {  // low word of (hi:lo) >> k, k in 0..31
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, k);
#else
  return k ? (lo >> k) | (hi << (32 - k)) : lo;
#endif
}
CS_HD M96 m_shr(M96 x, int k) {  // k in 0..31
  return m96(funnel_r(x.b, x.a, (unsigned)k), funnel_r(x.c, x.b, (unsigned)k), x.c >> k);
}
CS_HD M96 m_shl1(M96 x) { return m96(x.a << 1, (x.b << 1) | (x.a >> 31), (x.c << 1) | (x.b >> 31)); }
CS_HD M96 m_shr1(M96 x) { return m96((x.a >> 1) | (x.b << 31), (x.b >> 1) | (x.c << 31), x.c >> 1); }
CS_HD uint32_t word_below(int q) { return q >= 32 ? 0xFFFFFFFFu : (q <= 0 ? 0u : ~(0xFFFFFFFFu << q)); }
CS_HD M96 m_below(int q) { return m96(word_below(q), word_below(q - 32), word_below(q - 64)); }  // bits 0 .. q-1, q in 0..96
CS_HD M96 m_bit(int q) {
  const uint32_t bit = 1u << (q & 31);
  return m96(q < 32 ? bit : 0u, (q >= 32 && q < 64) ? bit : 0u, q >= 64 ? bit : 0u);
}
CS_HD unsigned m_test(M96 x, int q) { return ((q < 32 ? x.a : (q < 64 ? x.b : x.c)) >> (q & 31)) & 1u; }
CS_HD int m_ctz(M96 x) {  // x != 0
  return x.a ? __builtin_ctz(x.a) : (x.b ? 32 + __builtin_ctz(x.b) : 64 + __builtin_ctz(x.c));
}
CS_HD int m_top(M96 x) {  // x != 0: its highest set bit
  return x.c ? 95 - __builtin_clz(x.c) : (x.b ? 63 - __builtin_clz(x.b) : 31 - __builtin_clz(x.a));
}
CS_HD M96 m_from(U128 x) { return m96((uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.hi); }
CS_HD U128 m_to128(M96 x) { return cstd::u128(x.a | ((unsigned long long)x.b << 32), x.c); }

// All starts at which some alternative matches (`any`) and, for the span forms, the chosen alternative's length in binary
// across five planes.  `cls(k, off)` hands out the row's mask of class k shifted right by `off` (bit i = byte i + off is a
// member; cut at the row's length n <= kMaxRowBytes): the kernels cut it out of the class's bitmap at the shifted position,
// on demand -- a class mask is never held in registers beyond its use.  `raw(k, off)` is the same without the cut at the
// row's end (bits from n - off on are unspecified -- in the kernels they are the next row's): an alternative of length L
// only admits starts p with p + L <= n, so its items never look beyond the row and the one mask on the starts replaces a
// mask per item.
template <class Cls, class Raw>
CS_HD M96 starts(const View& V, Cls&& cls, Raw&& raw, int n, bool want_len, M96 plane[5]) {
  const M96 cursors = m_below(n + 1);
  M96 bnd = m96(0, 0, 0), bol = m96(1, 0, 0), eol = m_bit(n), bolm = bol, eolm = eol;
  if (V.flags & F_WORD) {
    const M96 W = cls(V.word_cls, 0);
    const M96 L = m_shl1(W);
    bnd = m_and(m96(W.a ^ L.a, W.b ^ L.b, W.c ^ L.c), cursors);
  }
  if (V.flags & (F_BOL_MULTI | F_EOL_MULTI)) {
    const M96 NL = cls(V.nl_cls, 0);
    bolm = m_or(bol, m_and(m_shl1(NL), cursors));  // behind a newline (regexec.inl: BOL with '^')
    eolm = m_or(eol, NL);                           // in front of a newline (EOL with '$')
  }
  M96 any = m96(0, 0, 0);
  if (want_len)
    for (int b = 0; b < 5; ++b) plane[b] = m96(0, 0, 0);
  const int32_t* w = V.img + kHeaderWords + kTableWords;
  for (int j = 0; j < V.J; ++j) {
    const int hdr = CSBITS_UNIFORM(*w++);
    const int items = hdr & 255, len = (hdr >> 8) & 255;
    M96 A = m_below(n - len + 1);  // starts whose match stays inside the row
    for (int i = 0; i < items; ++i) {
      const int it = CSBITS_UNIFORM(*w++);
      const int kind = it & 255, arg = (it >> 8) & 255, off = (it >> 16) & 255;
      if (kind == K_CLASS) {
        A = m_and(A, raw(arg, off));
        continue;
      }
      M96 X;
      switch (kind) {
        case K_BOW: X = bnd; break;
        case K_NBOW: X = m_andn(cursors, bnd); break;
        case K_BOL: X = bol; break;
        case K_EOL: X = eol; break;
        case K_BOL_MULTI: X = bolm; break;
        default: X = eolm; break;
      }
      A = m_and(A, m_shr(X, off));
    }
    if (want_len) {
      const M96 sel = m_andn(A, any);
      for (int b = 0; b < 5; ++b)
        if ((len >> b) & 1) plane[b] = m_or(plane[b], sel);
    }
    any = m_or(any, A);
  }
  return any;
}

// F_TAIL: the cursors at which every assertion behind the `+` loop holds
template <class Cls>
CS_HD M96 tail_cursors(const View& V, Cls&& cls, int n) {
  const M96 cursors = m_below(n + 1);
  M96 X = cursors;
  const uint32_t tw = (uint32_t)CSBITS_UNIFORM(V.img[CSBITS_UNIFORM(V.img[4]) - 1]);
  const int cnt = (int)(tw & 255u);
  for (int i = 0; i < cnt && i < 3; ++i) {
    const int kind = (int)((tw >> (8 * (i + 1))) & 255u);
    M96 A;
    if (kind == K_BOW || kind == K_NBOW) {
      const M96 W = cls(V.word_cls, 0);
      const M96 L = m_shl1(W);
      const M96 bnd = m_and(m96(W.a ^ L.a, W.b ^ L.b, W.c ^ L.c), cursors);
      A = kind == K_BOW ? bnd : m_andn(cursors, bnd);
    } else if (kind == K_BOL) {
      A = m96(1, 0, 0);
    } else if (kind == K_EOL) {
      A = m_bit(n);
    } else {
      const M96 NL = cls(V.nl_cls, 0);
      A = kind == K_BOL_MULTI ? m_or(m96(1, 0, 0), m_and(m_shl1(NL), cursors)) : m_or(m_bit(n), NL);
    }
    X = m_and(X, A);
  }
  return X;
}
// the cursor behind the match that starts with its fixed part ending at cursor `end` (F_PLUS): the end of the run of the last
// class -- with a tail the last cursor of the run at which the tail holds, -1 when there is none
// (`run_end`: where the run ends either way.  A start that fails takes with it every later start whose fixed part still ends
// inside the same run -- its exits are a subset: the callers drop the starts up to run_end - length at once, so a row costs a
// step per RUN, not per byte: `[^ ]+$` walked 55 failing starts a log line one by one, 44 ms on the C3 column)
CS_HD int plus_end(const View& V, M96 C, M96 X, int end, int& run_end) {
  // the run goes on from the match's last fixed byte: its end is the first non-member at or behind `end` (the row's end at
  // the latest: C is cut there, and n <= 95 leaves bit 95 clear)
  const M96 stop = m_andn(m96(~0u, ~0u, ~0u), m_or(C, m_below(end)));
  run_end = m_ctz(stop);
  if (!(V.flags & F_TAIL)) return run_end;
  const M96 cand = m_andn(m_and(X, m_below(run_end + 1)), m_below(end));
  return m_any(cand) ? m_top(cand) : -1;
}

// The row's matches in order, non-overlapping: S = first bytes, E = last bytes (one bit each per match).
template <class Cls, class Raw>
CS_HD void match(const View& V, Cls&& cls, Raw&& raw, int n, U128& S128, U128& E128) {
  if (V.flags & F_PURE_PLUS) {  // `[set]+` alone: the maximal runs of the class
    const M96 C = cls(V.plus_cls, 0);
    S128 = m_to128(m_andn(C, m_shl1(C)));
    E128 = m_to128(m_andn(C, m_shr1(C)));
    return;
  }
  if (V.flags & F_PURE_TAIL) {
    // `[set]+` and a tail: a run of the class holds at most one match -- from the run's first byte to the LAST cursor inside
    // the run at which the tail holds (what is left of the run behind it has no exit).  From the top: every tail cursor with a
    // member in front of it ends a match unless a higher one of its run did.
    const M96 C = cls(V.plus_cls, 0);
    M96 K = m_and(tail_cursors(V, cls, n), m_shl1(C));
    M96 S = m96(0, 0, 0), E = m96(0, 0, 0);
    while (m_any(K)) {
      const int q = m_top(K);
      const M96 out = m_andn(m_below(q), C);  // the non-members in front of q
      const int s0 = m_any(out) ? m_top(out) + 1 : 0;
      S = m_or(S, m_bit(s0));
      E = m_or(E, m_bit(q - 1));
      K = m_and(K, m_below(s0 + 1));
    }
    S128 = m_to128(S);
    E128 = m_to128(E);
    return;
  }
  M96 plane[5];
  const bool same_len = (V.flags & F_SAME_LEN) != 0;  // (one length for every alternative: no planes)
  const int the_len = CSBITS_UNIFORM(V.img[kHeaderWords + kTableWords]) >> 8 & 255;
  M96 rem = starts(V, cls, raw, n, !same_len, plane);
  M96 S = m96(0, 0, 0), E = m96(0, 0, 0), C = m96(0, 0, 0), X = m96(0, 0, 0);
  if (V.flags & F_PLUS) C = cls(V.plus_cls, 0);
  if (V.flags & F_TAIL) X = tail_cursors(V, cls, n);
  while (m_any(rem)) {
    const int p = m_ctz(rem);
    int len = the_len;
    if (!same_len) {
      len = 0;
      for (int b = 0; b < 5; ++b) len |= (int)m_test(plane[b], p) << b;
    }
    int end = p + len;  // the cursor behind the match
    if (V.flags & F_PLUS) {
      int run_end;
      end = plus_end(V, C, X, end, run_end);
      if (end < 0) {  // (no exit of the loop passes the tail: nothing matches at this start, nor at the starts that share its run)
        const int upto = run_end - len + 1;
        rem = m_andn(rem, m_below(upto > p + 1 ? upto : p + 1));
        continue;
      }
    }
    S = m_or(S, m_bit(p));
    E = m_or(E, m_bit(end - 1));
    rem = m_andn(rem, m_below(end));
  }
  S128 = m_to128(S);
  E128 = m_to128(E);
}

template <class Cls, class Raw>
CS_HD bool contains(const View& V, Cls&& cls, Raw&& raw, int n) {
  M96 unused[5];
  if (V.flags & F_PURE_TAIL) return m_any(m_and(tail_cursors(V, cls, n), m_shl1(cls(V.plus_cls, 0))));  // (a tail cursor with a member in front of it)
  if (V.flags & F_TAIL) {  // (a start counts only with an exit of its loop that passes the tail: one alternative, one length)
    M96 rem = starts(V, cls, raw, n, false, unused);
    const int len = CSBITS_UNIFORM(V.img[kHeaderWords + kTableWords]) >> 8 & 255;
    const M96 C = cls(V.plus_cls, 0), X = tail_cursors(V, cls, n);
    while (m_any(rem)) {
      const int p = m_ctz(rem);
      int run_end;
      if (plus_end(V, C, X, p + len, run_end) >= 0) return true;
      const int upto = run_end - len + 1;
      rem = m_andn(rem, m_below(upto > p + 1 ? upto : p + 1));
    }
    return false;
  }
  return m_any(starts(V, cls, raw, n, false, unused));
}
template <class Cls, class Raw>
CS_HD bool match_at_start(const View& V, Cls&& cls, Raw&& raw, int n) {
  M96 unused[5];
  const M96 st = starts(V, cls, raw, n, false, unused);
  if ((st.a & 1u) && (V.flags & F_TAIL)) {
    const int len = CSBITS_UNIFORM(V.img[kHeaderWords + kTableWords]) >> 8 & 255;
    int run_end;
    return plus_end(V, cls(V.plus_cls, 0), tail_cursors(V, cls, n), len, run_end) >= 0;
  }
  return (st.a & 1u) != 0;
}

}  // namespace csbits

namespace csrx {
struct Program;
// The bit program of `prog` (empty: the program does not convert).  `image` is the program's device image
// (Program::to_device_image), `flags` the 64 KiB unicode flag table.
std::vector<int32_t> build_bits(const Program& prog, const std::vector<int32_t>& image, const uint8_t* flags);
}  // namespace csrx
