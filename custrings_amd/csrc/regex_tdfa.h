// Tagged-DFA form of a compiled regex program: the ordered-thread (Pike)
// step function of regex_vm.h, precomputed on the host for every reachable
// (ordered thread list, previous-char category, seeding mode) and every input
// class, so that the device executes ONE table lookup per input byte instead of
// simulating thread lists.
//
// Exactness: a DFA state is the *ordered*, de-duplicated list of instructions
// the live threads are about to expand (priority = list order), so a transition
// reproduces regexec.inl:204-442 step for step -- including the cut of
// lower-priority threads on END and the stop of start-seeding after the first
// match.  Each transition also says where every surviving thread came from
// (an old slot or "new thread at this position"), which lets the executor carry
// the threads' START offsets in up to 4 registers: match spans come out
// identical to the list simulation.  Programs that need more than kMaxSlots
// simultaneous threads, more than kMaxStates states or more than 62 distinct
// character predicates are not converted (the caller keeps the list simulator).
//
// Image layout (int32 words):
//   [0] magic 'CSTD' [1] nstates [2] natoms [3] npreds [4] n_nonascii_atoms
//   [5] uses (bit0 word-category, bit1 line-category) [6] off INIT [7] off T1
//   [8] off T2 [9] off preds [10] off atomsig [11] off ACT [12] max slots
//   [13] min match chars [14] off CAT [15] total words
//   [16] nskip: states 0..nskip-1 are "idle" (no live thread, seeding on)
//   [17..20] idle state to continue in after skipping a byte of category c (0..3)
//   [21..24] candidate bitmap: ASCII bytes that must go through the table when idle
//   [25..28] word-character bitmap of the ASCII bytes (category bit 0)
//   [29] [30] two byte ranges (lo | hi << 8) that cover every candidate ASCII byte
//        (a superset is fine: extra candidates just take the table path)
//   [31] unit decomposition (regex_tdfa.cpp): bit 0 offered, bits 8..14 the byte x (0 = none), bit 16 = no match without an x
//   [29] bits 16..31, [30] bits 16..19: the CHAIN form of the pattern (regex_tdfa.cpp; chain_match below): up to eight items,
//        two bits each (bit 0: the item's class is the byte x instead of the candidate ranges, bit 1: repeated, `+`), and
//        their number (0 = the pattern is no chain); [30] bits 21..23: the chain is followed by that many literal bytes (the
//        SUFFIX, none of them in the candidate ranges; the word in front of the image's last one -- or the last one when no
//        group map follows -- holds them, first byte lowest): `(\d+)\.(\d+)\.\d+\.(\d+) `; [30] bits 24 / 25: a `\b` in front of
//        the chain / behind it, bit 26: the general form of the arithmetic (a counted item or a `\\b`: chain_match_counted).  Two words in front of those tail words: how often each item is taken, a byte per item (low
//        nibble the least, high nibble the most repetitions, 0 = unbounded): `\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b`
//   INIT   : 3 modes x 8 categories state ids
//   T1     : nstates x 128 entries (ASCII byte -> transition; byte 0 = embedded NUL)
//   T2     : nstates x natoms entries (atom 0 = end of row, 1 = embedded NUL,
//            2.. = character classes; used for non-ASCII characters)
//   CAT    : 128 bytes, category bits of each ASCII byte (bit0 word, bit1 newline)
//   preds  : npreds x {type, arg}
//   atomsig: n_nonascii_atoms x {sig_lo, sig_hi, atom}
//   ACT    : origin words of the "complex" transitions
// Transition entry (uint32):
//   [9:0] next state  [10] STOP  [11] MATCH  [15:12] match origin (0-7 slot, 15 = new)
//   [19:16] keep ^ 15 (slots j >= keep start at this position; keep 15 = unchanged), so
//           that a transition with no side effect at all is just its next-state id (< 1024)
//   [20] COMPLEX (origins in ACT[entry >> 21], 4 bits per slot, 15 = new)
//   [21] (entries without COMPLEX only) EXIT: the next state is an idle state
// (slot fields are 4 bits wide although kMaxSlots is 4: room to grow)
#pragma once
#include <stdint.h>

#include <vector>

#include "regex_vm.h"

namespace cstd {

constexpr int32_t kMagic = 0x44545343;  // "CSTD"
constexpr int kMaxSlots = 4;
// Programs that keep five to eight threads alive (counted repetitions: \d{1,3}\.\d{1,3}..., \w{5}, [0-9a-f]{8}-...) convert
// too; they run on the generic executor with eight start offsets (TdfaWide below) -- not on the lean scans, the unit
// decomposition or the capture-group runs, all of which pack four slots into one register.
constexpr int kMaxSlotsWide = 8;
constexpr int kMaxStates = 512;
constexpr int kHeaderWords = 32;
enum { MODE_RESTART = 0, MODE_NORESTART = 1, MODE_SEED_ONCE = 2 };
enum { ATOM_EOT = 0, ATOM_NUL = 1, ATOM_FIRST_CLASS = 2 };
enum { P_CHAR = 0, P_ANY = 1, P_ANYNL = 2, P_CCLASS = 3, P_NCCLASS = 4, P_ISNL = 5, P_ISWORD = 6 };

constexpr uint32_t E_STATE = 0x3FFu, E_STOP = 1u << 10, E_MATCH = 1u << 11, E_COMPLEX = 1u << 20;
// bit 21 of an entry WITHOUT E_COMPLEX (whose bits 31:21 are otherwise zero): the next state is
// an idle one -- lets the lean scan test "leave the inner loop" with a single mask
constexpr uint32_t E_EXIT = 1u << 21;
CS_HD uint32_t e_match_origin(uint32_t e) { return (e >> 12) & 15u; }
CS_HD uint32_t e_keep(uint32_t e) { return ((e >> 16) & 15u) ^ 15u; }
CS_HD uint32_t e_keep_field(uint32_t keep) { return ((keep ^ 15u) & 15u) << 16; }
constexpr uint32_t E_ACTION = E_MATCH | E_COMPLEX | (15u << 16);  // any bit set / keep != 15 handled below

// 128-bit masks for the unit decomposition (rows of up to 96 bytes; bit 96 may hold a carry)
struct U128 {
  unsigned long long lo, hi;
};
CS_HD U128 u128(unsigned long long lo, unsigned long long hi) {
  U128 r;
  r.lo = lo;
  r.hi = hi;
  return r;
}
CS_HD U128 u128_add(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}
CS_HD U128 u128_sub(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo - b.lo;
  r.hi = a.hi - b.hi - (a.lo < b.lo ? 1ull : 0ull);
  return r;
}
CS_HD U128 u128_shl1(U128 a) { return u128(a.lo << 1, (a.hi << 1) | (a.lo >> 63)); }
CS_HD U128 u128_andn(U128 a, U128 b) { return u128(a.lo & ~b.lo, a.hi & ~b.hi); }  // a & ~b
CS_HD U128 u128_and(U128 a, U128 b) { return u128(a.lo & b.lo, a.hi & b.hi); }
CS_HD U128 u128_or(U128 a, U128 b) { return u128(a.lo | b.lo, a.hi | b.hi); }
CS_HD bool u128_any(U128 a) { return (a.lo | a.hi) != 0; }
CS_HD int u128_popc(U128 a) { return __builtin_popcountll(a.lo) + __builtin_popcountll(a.hi); }
CS_HD int u128_ctz(U128 a) { return a.lo ? __builtin_ctzll(a.lo) : 64 + __builtin_ctzll(a.hi); }  // a != 0
CS_HD int u128_msb(U128 a) { return a.hi ? 127 - __builtin_clzll(a.hi) : (a.lo ? 63 - __builtin_clzll(a.lo) : -1); }
CS_HD U128 u128_clear_lowest(U128 a) { return a.lo ? u128(a.lo & (a.lo - 1), a.hi) : u128(0, a.hi & (a.hi - 1)); }
CS_HD U128 u128_below(int q) {  // bits 0 .. q-1, q in 0..128
  if (q >= 128) return u128(~0ull, ~0ull);
  if (q >= 64) return u128(~0ull, q == 64 ? 0ull : ~(~0ull << (q - 64)));
  return u128(q == 0 ? 0ull : ~(~0ull << q), 0ull);
}
// Units of a row (see regex_tdfa.cpp): C = candidate bits, X = "byte equals x" bits, both row-relative and cut at
// the row length.  Returns one bit per unit, at the position just behind the unit's last byte; N receives C | X.
// (Adding a subset M of N to N carries out of exactly those runs of N that hold a bit of M, into the zero above them.)
CS_HD U128 unit_ends(U128 C, U128 X, bool required, U128& N) {
  N = u128_or(C, X);
  U128 w = u128_andn(u128_add(N, C), N);
  if (required) w = u128_and(w, u128_andn(u128_add(N, X), N));
  return w;
}
// first byte of the run of N that ends just below bit q
CS_HD int unit_start(U128 N, int q) { return u128_msb(u128_andn(u128_below(q), N)) + 1; }

// ---- chain patterns: every match of a row by bit arithmetic on two per-byte masks, no automaton ----
// A CHAIN is a sequence of up to eight items, each one byte class taken a counted number of times -- once, `+` (greedy), or
// `{m,n}` / `{m,}` / `{m}` with n <= 15 (crep: a byte per item, low nibble the least and high nibble the most repetitions, 0 =
// unbounded) -- where the class is either R (exactly the ASCII bytes of the two candidate ranges) or the single byte x,
// neighbours differ, and the last item is repeated when its class is the first one's: `\d+\.\d+\.\d+\.\d+`, `[a-z]+=`,
// `\d+`.  A `\b` may stand in front of the chain and behind it when the class next to it holds letters and digits only
// (chain bits 24 / 25): `\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b`.  On a row of plain ASCII such a
// pattern is deterministic: from a given start there is at most ONE match (a repeated item must take its whole run, the
// next item's class being disjoint -- which is also why a bounded item is a run of at most n bytes and nothing subtler; a
// bounded FIRST item needs the `\b` in front of it, or the match could begin inside the run, and a bounded last one a `\b`
// or a suffix behind it: regex_tdfa.cpp checks both), a match exists from the middle of the first run only if it exists from
// the run's start, and a later start ends later.  So (Parabix-style marker arithmetic, one marker set for ALL starts of the
// row at once):
//   forward   M = run starts of R (every R byte when the first item is single); per item M = (M & C) << 1 for each
//             repetition it must have, then for an unbounded one M = MatchStar(M, C) = (((M & C) + C) ^ C) | M, for a bounded
//             one M |= (M & C) << 1 per optional repetition; the ends are what is left (off C for a repeated tail)
//   backward  the same walk over the bit-reversed masks from the ends: the starts that do reach an end
//   pairing   the k-th start belongs to the k-th end; a match that begins inside the previous one is dropped (the scan
//             of regexec.inl:204-442 resumes at the end of a match -- and that position is never inside a first run), and
//             so is one whose start is no word boundary when the pattern asks for one.
// R, X: row-relative bits cut at the row length (rows of up to 96 bytes).  S / L receive a bit per match at its first /
// last byte.  ~300 integer operations a row whatever it holds, against a table walk per candidate byte.
#if !defined(__HIP_DEVICE_COMPILE__)
inline int g_chain_host = 1;  // host builds (tests/rowemu): 0 keeps chain patterns on the unit route, 2 runs the two-half form of the arithmetic, so that all are checked
#endif
CS_HD U128 u128_xor(U128 a, U128 b) { return u128(a.lo ^ b.lo, a.hi ^ b.hi); }
CS_HD U128 u128_shr1(U128 a) { return u128((a.lo >> 1) | (a.hi << 63), a.hi >> 1); }
CS_HD unsigned long long u64_bitrev(unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brevll(v);
#else
  v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
  v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
  v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
  return __builtin_bswap64(v);
#endif
}
// bit p -> bit 126 - p (p <= 126; its own inverse)
CS_HD U128 u128_rev127(U128 a) { return u128_shr1(u128(u64_bitrev(a.hi), u64_bitrev(a.lo))); }
CS_HD U128 chain_star(U128 M, U128 C) { return u128_or(u128_xor(u128_add(u128_and(M, C), C), C), M); }
// The chain's description is the same in every lane, but it comes out of the image in LDS, i.e. in vector registers: taken
// into scalar registers once per call, the item loops below branch and shift on the scalar unit (without this the counted
// form cost the headline's replace_re kernel 0.55 ms of 4.7: vector compares and exec-masked loops per item).
#if defined(__HIP_DEVICE_COMPILE__)
#define CS_CHAIN_UNIFORM(chain, crep)                                                                                      \
  do {                                                                                                                     \
    chain = (uint32_t)__builtin_amdgcn_readfirstlane((int)(chain));                                                        \
    crep = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(crep)) |                            \
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((crep) >> 32)) << 32);            \
  } while (0)
#else
#define CS_CHAIN_UNIFORM(chain, crep) \
  do {                                \
  } while (0)
#endif
constexpr uint32_t kChainLeadB = 1u << 24, kChainTrailB = 1u << 25;  // `\b` in front of the chain / behind it
// the bytes `\b` counts as word characters on a plain-ASCII row (regexec.inl:290-299: alphanumeric; '_' is none)
CS_HD bool chain_word_byte(uint32_t c) { return c - 48u < 10u || (c | 32u) - 97u < 26u; }
CS_HD uint32_t chain_rep(unsigned long long crep, int k) { return (uint32_t)(crep >> (8 * k)) & 255u; }
// is item k a run (it takes every byte of its class that is there), as against a single byte?
CS_HD bool chain_runs(unsigned long long crep, int k) { return (chain_rep(crep, k) >> 4) != 1u; }
// the markers behind one item: M in front of it, C its class
CS_HD U128 chain_item(U128 M, U128 C, uint32_t rep) {
  M = u128_shl1(u128_and(M, C));  // (every item is taken once at least)
  // (the two common items first, a uniform branch each: the counting loops below cost the dotted quad of the headline a
  // tenth of its replace_re kernel when every item went through them)
  if (rep == 0x11u) return M;
  if (rep == 0x01u) return chain_star(M, C);
  const int least = (int)(rep & 15u), most = (int)(rep >> 4);
  for (int i = 1; i < least; ++i) M = u128_shl1(u128_and(M, C));
  if (most == 0) return chain_star(M, C);
  U128 T = M;
  for (int i = least; i < most; ++i) {
    T = u128_shl1(u128_and(T, C));
    M = u128_or(M, T);
  }
  return M;
}
// the last byte of every match of the row, whatever its start (a `\b` in front of the chain is NOT looked at: chain_match)
CS_HD U128 chain_ends_counted(U128 R, U128 X, uint32_t chain, unsigned long long crep) {
  CS_CHAIN_UNIFORM(chain, crep);
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  U128 M = chain_runs(crep, 0) ? u128_andn(R, u128_shl1(R)) : R;
  U128 C = R;
  for (int k = 0; k < ni; ++k) {
    C = is_x(k) ? X : R;
    M = chain_item(M, C, chain_rep(crep, k));
  }
  if (chain_runs(crep, ni - 1)) M = u128_andn(M, C);
  return u128_shr1(M);
}
// The chain's SUFFIX: literal bytes behind the last item (none in R, so a repeated last item still takes its whole run and
// a start has one match or none), or a `\b` there.  Of the ends of the chain part only those stay that the suffix follows;
// byte_at(i) is byte i of the row (n bytes).
template <class ByteAt>
CS_HD U128 chain_suffix_filter(U128 Le, uint32_t chain, uint32_t sfx, int n, ByteAt&& byte_at) {
#if defined(__HIP_DEVICE_COMPILE__)
  chain = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain);
  sfx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sfx);
#endif
  const int sl = (int)((chain >> 20) & 7u);
  const bool wb = (chain & kChainTrailB) != 0;
  if (!sl && !wb) return Le;
  U128 T = Le;
  while (u128_any(T)) {
    const int l = u128_ctz(T);
    const U128 rest = u128_clear_lowest(T);
    bool ok = l + 1 + sl <= n;
    for (int k = 0; k < sl && ok; ++k) ok = (uint32_t)byte_at(l + 1 + k) == ((sfx >> (8 * k)) & 255u);
    if (wb && l + 1 < n) ok = !chain_word_byte((uint32_t)byte_at(l + 1));  // (the last byte is a word character: regex_tdfa.cpp)
    if (!ok) Le = u128_andn(Le, u128_andn(T, rest));
    T = rest;
  }
  return Le;
}
template <class ByteAt>
CS_HD void chain_match_counted(U128 R, U128 X, uint32_t chain, unsigned long long crep, U128& S, U128& L, uint32_t sfx, int n, ByteAt&& byte_at) {
  CS_CHAIN_UNIFORM(chain, crep);
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  U128 Le = chain_suffix_filter(chain_ends_counted(R, X, chain, crep), chain, sfx, n, byte_at);
  S = u128(0, 0);
  L = u128(0, 0);
  if (!u128_any(Le)) return;
  U128 M, C = R;
  // backward: the reversed chain over the reversed masks, from the ends
  const U128 Rr = u128_rev127(R), Xr = u128_rev127(X);
  M = u128_rev127(Le);
  for (int k = ni - 1; k >= 0; --k) {
    C = is_x(k) ? Xr : Rr;
    M = chain_item(M, C, chain_rep(crep, k));
  }
  if (chain_runs(crep, 0)) M = u128_andn(M, C);
  U128 Sv = u128_rev127(u128_shr1(M));  // the starts that reach an end, as many as there are ends
  {
    // (the match's last byte is the suffix's: rows end within 96 bytes, the shift loses nothing)
    const int sl = (int)((chain >> 20) & 7u);
    if (sl) Le = u128(Le.lo << sl, (Le.hi << sl) | (Le.lo >> (64 - sl)));
  }
  const bool wb = (chain & kChainLeadB) != 0;
  int cursor = 0;
  while (u128_any(Sv) && u128_any(Le)) {
    const int s = u128_ctz(Sv), l = u128_ctz(Le);
    const U128 sb = u128_andn(Sv, u128_clear_lowest(Sv)), lb = u128_andn(Le, u128_clear_lowest(Le));
    Sv = u128_clear_lowest(Sv);
    Le = u128_clear_lowest(Le);
    // (`\b` in front: the first byte is a word character, so the one before it must be none)
    if (s >= cursor && !(wb && s > 0 && chain_word_byte((uint32_t)byte_at(s - 1)))) {
      S = u128_or(S, sb);
      L = u128_or(L, lb);
      cursor = l + 1;
    }
  }
}

// The capture groups of a chain match (header word 30 bit 20; gmap: a byte per group 1..4, low nibble the group's first
// item, high nibble the item behind its last): every group is a run of items, so its range follows from the item
// boundaries of the match that starts at mb -- a walk over the row's two masks, no automaton.  gb / ge: -1 for a group
// the map does not name.
CS_HD void chain_group_bounds_counted(U128 R, U128 X, uint32_t chain, unsigned long long crep, uint32_t gmap, int mb, int gb[4], int ge[4]) {
  CS_CHAIN_UNIFORM(chain, crep);
#if defined(__HIP_DEVICE_COMPILE__)
  gmap = (uint32_t)__builtin_amdgcn_readfirstlane((int)gmap);
#endif
  const int ni = (int)((chain >> 16) & 15u);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 0; q < 4; ++q) gb[q] = ge[q] = -1;
  int p = mb;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k <= 8; ++k) {
    if (k <= ni) {  // (uniform) p = the boundary in front of item k
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int q = 0; q < 4; ++q) {
        if ((int)((gmap >> (8 * q)) & 15u) == k) gb[q] = p;
        if ((int)((gmap >> (8 * q + 4)) & 15u) == k) ge[q] = p;
      }
    }
    if (k < ni && k < 8) {
      if (chain_runs(crep, k)) {
        const U128 C = ((chain >> (2 * k)) & 1u) ? X : R;
        p = u128_ctz(u128_andn(u128(~C.lo, ~C.hi), u128_below(p)));  // the first byte at or behind p off the class
      } else {
        ++p;
      }
    }
  }
}

// ---- the same for chains of single and `+` items alone, as they were before the counted form: the headline's dotted quad runs
// ---- these (the general form above cost its replace_re kernel 0.6 of 4.7 ms even with the fast paths in chain_item)
// the last byte of every match of the row, whatever its start (contains_re needs no more than "any")
CS_HD U128 chain_ends_plain(U128 R, U128 X, uint32_t chain) {
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  auto plus = [&](int k) { return ((chain >> (2 * k + 1)) & 1u) != 0; };
  U128 M = plus(0) ? u128_andn(R, u128_shl1(R)) : R;
  U128 C = R;
  for (int k = 0; k < ni; ++k) {
    C = is_x(k) ? X : R;
    M = u128_shl1(u128_and(M, C));
    if (plus(k)) M = chain_star(M, C);
  }
  if (plus(ni - 1)) M = u128_andn(M, C);
  return u128_shr1(M);
}
// The chain's SUFFIX: literal bytes behind the last item (none in R, so a repeated last item still takes its whole run and
// a start has one match or none).  Of the ends of the chain part only those stay that the suffix follows; byte_at(i) is
// byte i of the row (n bytes).
template <class ByteAt>
CS_HD U128 chain_suffix_filter_plain(U128 Le, uint32_t chain, uint32_t sfx, int n, ByteAt&& byte_at) {
  const int sl = (int)((chain >> 20) & 7u);
  if (!sl) return Le;
  U128 T = Le;
  while (u128_any(T)) {
    const int l = u128_ctz(T);
    const U128 rest = u128_clear_lowest(T);
    bool ok = l + 1 + sl <= n;
    for (int k = 0; k < sl && ok; ++k) ok = (uint32_t)byte_at(l + 1 + k) == ((sfx >> (8 * k)) & 255u);
    if (!ok) Le = u128_andn(Le, u128_andn(T, rest));
    T = rest;
  }
  return Le;
}
template <class ByteAt>
CS_HD void chain_match_plain(U128 R, U128 X, uint32_t chain, U128& S, U128& L, uint32_t sfx, int n, ByteAt&& byte_at) {
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  auto plus = [&](int k) { return ((chain >> (2 * k + 1)) & 1u) != 0; };
  U128 Le = chain_suffix_filter_plain(chain_ends_plain(R, X, chain), chain, sfx, n, byte_at);
  S = u128(0, 0);
  L = u128(0, 0);
  if (!u128_any(Le)) return;
  U128 M, C = R;
  // backward: the reversed chain over the reversed masks, from the ends
  const U128 Rr = u128_rev127(R), Xr = u128_rev127(X);
  M = u128_rev127(Le);
  for (int k = ni - 1; k >= 0; --k) {
    C = is_x(k) ? Xr : Rr;
    M = u128_shl1(u128_and(M, C));
    if (plus(k)) M = chain_star(M, C);
  }
  if (plus(0)) M = u128_andn(M, C);
  U128 Sv = u128_rev127(u128_shr1(M));  // the starts that reach an end, as many as there are ends
  {
    // (the match's last byte is the suffix's: rows end within 96 bytes, the shift loses nothing)
    const int sl = (int)((chain >> 20) & 7u);
    if (sl) Le = u128(Le.lo << sl, (Le.hi << sl) | (Le.lo >> (64 - sl)));
  }
  int cursor = 0;
  while (u128_any(Sv) && u128_any(Le)) {
    const int s = u128_ctz(Sv), l = u128_ctz(Le);
    const U128 sb = u128_andn(Sv, u128_clear_lowest(Sv)), lb = u128_andn(Le, u128_clear_lowest(Le));
    Sv = u128_clear_lowest(Sv);
    Le = u128_clear_lowest(Le);
    if (s >= cursor) {
      S = u128_or(S, sb);
      L = u128_or(L, lb);
      cursor = l + 1;
    }
  }
}

// The capture groups of a chain match (header word 30 bit 20; gmap: a byte per group 1..4, low nibble the group's first
// item, high nibble the item behind its last): every group is a run of items, so its range follows from the item
// boundaries of the match that starts at mb -- a walk over the row's two masks, no automaton.  gb / ge: -1 for a group
// the map does not name.
CS_HD void chain_group_bounds_plain(U128 R, U128 X, uint32_t chain, uint32_t gmap, int mb, int gb[4], int ge[4]) {
  const int ni = (int)((chain >> 16) & 15u);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 0; q < 4; ++q) gb[q] = ge[q] = -1;
  int p = mb;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k <= 8; ++k) {
    if (k <= ni) {  // (uniform) p = the boundary in front of item k
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int q = 0; q < 4; ++q) {
        if ((int)((gmap >> (8 * q)) & 15u) == k) gb[q] = p;
        if ((int)((gmap >> (8 * q + 4)) & 15u) == k) ge[q] = p;
      }
    }
    if (k < ni && k < 8) {
      if ((chain >> (2 * k + 1)) & 1u) {
        const U128 C = ((chain >> (2 * k)) & 1u) ? X : R;
        p = u128_ctz(u128_andn(u128(~C.lo, ~C.hi), u128_below(p)));  // the first byte at or behind p off the class
      } else {
        ++p;
      }
    }
  }
}


// ---- the plain form on THREE 32-bit words (rows of at most 95 bytes: the kernels take these forms up to 93): a quarter fewer
// ---- integer operations than on two 64-bit halves -- every 64-bit operation is two 32-bit ones on this machine, the fourth word
// ---- was always zero.  Positions reverse as p -> 94 - p (the backward walk ends one beyond the start: bit 95 for a start at 0).
struct W96 {
  uint32_t a, b, c;
};
CS_HD W96 w96(uint32_t a, uint32_t b, uint32_t c) {
  W96 r;
  r.a = a;
  r.b = b;
  r.c = c;
  return r;
}
CS_HD W96 w_and(W96 x, W96 y) { return w96(x.a & y.a, x.b & y.b, x.c & y.c); }
CS_HD W96 w_or(W96 x, W96 y) { return w96(x.a | y.a, x.b | y.b, x.c | y.c); }
CS_HD W96 w_xor(W96 x, W96 y) { return w96(x.a ^ y.a, x.b ^ y.b, x.c ^ y.c); }
CS_HD W96 w_andn(W96 x, W96 y) { return w96(x.a & ~y.a, x.b & ~y.b, x.c & ~y.c); }  // x & ~y
CS_HD bool w_any(W96 x) { return (x.a | x.b | x.c) != 0; }
CS_HD W96 w_shl1(W96 x) { return w96(x.a << 1, (x.b << 1) | (x.a >> 31), (x.c << 1) | (x.b >> 31)); }
CS_HD W96 w_shr1(W96 x) { return w96((x.a >> 1) | (x.b << 31), (x.b >> 1) | (x.c << 31), x.c >> 1); }
CS_HD W96 w_add(W96 x, W96 y) {
  unsigned long long t = (unsigned long long)x.a + y.a;
  const uint32_t a = (uint32_t)t;
  t = (unsigned long long)x.b + y.b + (t >> 32);
  return w96(a, (uint32_t)t, x.c + y.c + (uint32_t)(t >> 32));
}
CS_HD uint32_t u32_bitrev(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(v);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(v);
#endif
}
CS_HD W96 w_rev94(W96 x) { return w_shr1(w96(u32_bitrev(x.c), u32_bitrev(x.b), u32_bitrev(x.a))); }  // bit p -> bit 94 - p (its own inverse)
CS_HD int w_ctz(W96 x) { return x.a ? __builtin_ctz(x.a) : (x.b ? 32 + __builtin_ctz(x.b) : 64 + __builtin_ctz(x.c)); }  // x != 0
CS_HD W96 w_clear_lowest(W96 x) {
  // x & (x - 1)
  const uint32_t a = x.a - 1u, b = x.b - (x.a == 0u ? 1u : 0u), c = x.c - ((x.a | x.b) == 0u ? 1u : 0u);
  return w96(x.a & a, x.b & b, x.c & c);
}
CS_HD W96 w_star(W96 M, W96 C) { return w_or(w_xor(w_add(w_and(M, C), C), C), M); }
CS_HD U128 w_to128(W96 x) { return u128(x.a | ((unsigned long long)x.b << 32), x.c); }
// chain_match_plain on 96-bit words: the same walks, the same pairing (rows of at most 95 bytes)
template <class ByteAt>
CS_HD void chain_match_plain96(W96 R, W96 X, uint32_t chain, U128& S128, U128& L128, uint32_t sfx, int n, ByteAt&& byte_at) {
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  auto plus = [&](int k) { return ((chain >> (2 * k + 1)) & 1u) != 0; };
  W96 M = plus(0) ? w_andn(R, w_shl1(R)) : R, C = R;
  for (int k = 0; k < ni; ++k) {
    C = is_x(k) ? X : R;
    M = w_shl1(w_and(M, C));
    if (plus(k)) M = w_star(M, C);
  }
  if (plus(ni - 1)) M = w_andn(M, C);
  W96 Le = w_shr1(M);
  const int sl = (int)((chain >> 20) & 7u);
  if (sl) {  // the suffix: of the ends only those it follows
    W96 T = Le;
    while (w_any(T)) {
      const int l = w_ctz(T);
      const W96 rest = w_clear_lowest(T);
      bool ok = l + 1 + sl <= n;
      for (int k = 0; k < sl && ok; ++k) ok = (uint32_t)byte_at(l + 1 + k) == ((sfx >> (8 * k)) & 255u);
      if (!ok) Le = w_andn(Le, w_andn(T, rest));
      T = rest;
    }
  }
  W96 S = w96(0, 0, 0), L = w96(0, 0, 0);
  if (w_any(Le)) {
    const W96 Rr = w_rev94(R), Xr = w_rev94(X);
    M = w_rev94(Le);
    for (int k = ni - 1; k >= 0; --k) {
      C = is_x(k) ? Xr : Rr;
      M = w_shl1(w_and(M, C));
      if (plus(k)) M = w_star(M, C);
    }
    if (plus(0)) M = w_andn(M, C);
    W96 Sv = w_rev94(w_shr1(M));
    if (sl) {  // (the match's last byte is the suffix's; rows end within 95 bytes, the shift loses nothing)
      const unsigned long long lo = (unsigned long long)Le.a | ((unsigned long long)Le.b << 32);
      Le = w96((uint32_t)(lo << sl), (uint32_t)((lo << sl) >> 32), (Le.c << sl) | (uint32_t)(lo >> (64 - sl)));
    }
    int cursor = 0;
    while (w_any(Sv) && w_any(Le)) {
      const int s = w_ctz(Sv), l = w_ctz(Le);
      const W96 sr = w_clear_lowest(Sv), lr = w_clear_lowest(Le);
      if (s >= cursor) {
        S = w_or(S, w_andn(Sv, sr));
        L = w_or(L, w_andn(Le, lr));
        cursor = l + 1;
      }
      Sv = sr;
      Le = lr;
    }
  }
  S128 = w_to128(S);
  L128 = w_to128(L);
}
// ... and the general form (counted items, `\\b`) on the same words
CS_HD W96 chain_item96(W96 M, W96 C, uint32_t rep) {
  M = w_shl1(w_and(M, C));
  if (rep == 0x11u) return M;
  if (rep == 0x01u) return w_star(M, C);
  const int least = (int)(rep & 15u), most = (int)(rep >> 4);
  for (int i = 1; i < least; ++i) M = w_shl1(w_and(M, C));
  if (most == 0) return w_star(M, C);
  W96 T = M;
  for (int i = least; i < most; ++i) {
    T = w_shl1(w_and(T, C));
    M = w_or(M, T);
  }
  return M;
}
template <class ByteAt>
CS_HD void chain_match_counted96(W96 R, W96 X, uint32_t chain, unsigned long long crep, U128& S128, U128& L128, uint32_t sfx, int n, ByteAt&& byte_at) {
  CS_CHAIN_UNIFORM(chain, crep);
  const int ni = (int)((chain >> 16) & 15u);
  auto is_x = [&](int k) { return ((chain >> (2 * k)) & 1u) != 0; };
  W96 M = chain_runs(crep, 0) ? w_andn(R, w_shl1(R)) : R, C = R;
  for (int k = 0; k < ni; ++k) {
    C = is_x(k) ? X : R;
    M = chain_item96(M, C, chain_rep(crep, k));
  }
  if (chain_runs(crep, ni - 1)) M = w_andn(M, C);
  W96 Le = w_shr1(M);
  const int sl = (int)((chain >> 20) & 7u);
  const bool tb = (chain & kChainTrailB) != 0, lb = (chain & kChainLeadB) != 0;
  if (sl || tb) {  // the suffix / the closing `\\b`: of the ends only those it follows
    W96 T = Le;
    while (w_any(T)) {
      const int l = w_ctz(T);
      const W96 rest = w_clear_lowest(T);
      bool ok = l + 1 + sl <= n;
      for (int k = 0; k < sl && ok; ++k) ok = (uint32_t)byte_at(l + 1 + k) == ((sfx >> (8 * k)) & 255u);
      if (tb && l + 1 < n) ok = !chain_word_byte((uint32_t)byte_at(l + 1));
      if (!ok) Le = w_andn(Le, w_andn(T, rest));
      T = rest;
    }
  }
  W96 S = w96(0, 0, 0), L = w96(0, 0, 0);
  if (w_any(Le)) {
    const W96 Rr = w_rev94(R), Xr = w_rev94(X);
    M = w_rev94(Le);
    for (int k = ni - 1; k >= 0; --k) {
      C = is_x(k) ? Xr : Rr;
      M = chain_item96(M, C, chain_rep(crep, k));
    }
    if (chain_runs(crep, 0)) M = w_andn(M, C);
    W96 Sv = w_rev94(w_shr1(M));
    if (sl) {
      const unsigned long long lo = (unsigned long long)Le.a | ((unsigned long long)Le.b << 32);
      Le = w96((uint32_t)(lo << sl), (uint32_t)((lo << sl) >> 32), (Le.c << sl) | (uint32_t)(lo >> (64 - sl)));
    }
    int cursor = 0;
    while (w_any(Sv) && w_any(Le)) {
      const int s = w_ctz(Sv), l = w_ctz(Le);
      const W96 sr = w_clear_lowest(Sv), lr = w_clear_lowest(Le);
      if (s >= cursor && !(lb && s > 0 && chain_word_byte((uint32_t)byte_at(s - 1)))) {
        S = w_or(S, w_andn(Sv, sr));
        L = w_or(L, w_andn(Le, lr));
        cursor = l + 1;
      }
      Sv = sr;
      Le = lr;
    }
  }
  S128 = w_to128(S);
  L128 = w_to128(L);
}
// the row's masks as they come out of the bitmaps (three words each)
template <class ByteAt>
CS_HD void chain_match96(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t chain, const int32_t* img, U128& S, U128& L,
                         uint32_t sfx, int n, ByteAt&& byte_at);

constexpr uint32_t kChainCounted = 1u << 23;  // some item is counted, or a `\\b` stands at an end: the general form
// the items' repetition counts (chain_item), a byte each: the two words in front of the suffix / group-map words at the image's
// end (read where the general form runs, not kept in the view: the plain form's kernels do not pay registers for them)
CS_HD unsigned long long chain_crep(const int32_t* img) {
  const int at = img[15] - 2 - (int)(((uint32_t)img[30] >> 20) & 1u) - ((((uint32_t)img[30] >> 21) & 7u) ? 1 : 0);
  return (unsigned long long)(uint32_t)img[at] | ((unsigned long long)(uint32_t)img[at + 1] << 32);
}
CS_HD U128 chain_ends(U128 R, U128 X, uint32_t chain, const int32_t* img) {
  if (chain & kChainCounted) return chain_ends_counted(R, X, chain, chain_crep(img));
  return chain_ends_plain(R, X, chain);
}
template <class ByteAt>
CS_HD void chain_match(U128 R, U128 X, uint32_t chain, const int32_t* img, U128& S, U128& L, uint32_t sfx, int n, ByteAt&& byte_at) {
  if (chain & kChainCounted) chain_match_counted(R, X, chain, chain_crep(img), S, L, sfx, n, byte_at);
  else chain_match_plain(R, X, chain, S, L, sfx, n, byte_at);
}
template <class ByteAt>
CS_HD void chain_match96(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t chain, const int32_t* img, U128& S, U128& L,
                         uint32_t sfx, int n, ByteAt&& byte_at) {
  if (chain & kChainCounted) chain_match_counted96(w96(r0, r1, r2), w96(x0, x1, x2), chain, chain_crep(img), S, L, sfx, n, byte_at);
  else chain_match_plain96(w96(r0, r1, r2), w96(x0, x1, x2), chain, S, L, sfx, n, byte_at);
}
CS_HD void chain_group_bounds(U128 R, U128 X, uint32_t chain, const int32_t* img, uint32_t gmap, int mb, int gb[4], int ge[4]) {
  if (chain & kChainCounted) chain_group_bounds_counted(R, X, chain, chain_crep(img), gmap, mb, gb, ge);
  else chain_group_bounds_plain(R, X, chain, gmap, mb, gb, ge);
}

struct View {
  const int32_t* img;
  const uint32_t* init;
  const uint32_t* t1;
  const uint32_t* t2;
  const uint32_t* cat;  // 32 words = 128 bytes
  const int32_t* preds;
  const uint32_t* atomsig;
  const uint32_t* act;
  int nstates, natoms, npreds, nna, uses;
  uint32_t nskip;
  // scalars, not arrays: an array member would force the whole view into memory
  uint32_t r1lo, r1hi, r2lo, r2hi;  // SWAR constants of the two candidate ranges
  uint32_t skippack, cand0, cand1, cand2, cand3, word0, word1, word2, word3;
  uint32_t units;  // header word 31
  uint32_t chain;  // bits 0..15 the items, bits 16..19 their number (header words 29 / 30, upper halves), bits 20..22 suffix bytes, 24 / 25 `\\b` in front / behind
  uint32_t sfx;    // the suffix bytes
};
// (`tables`: where the tables lie when `img` is only a copy of the header and of the image's last words -- cs_regex.hip,
// tsetup: the forms that leave the tables in memory keep those 40 words in LDS, kHeadTailWords)
constexpr int kHeadTailWords = 40;  // the 32 header words and the last 8 of the image (word 15 of the copy says 40)
CS_HD View make_view(const int32_t* img, const int32_t* tables = nullptr) {
  View v;
  v.img = img;
  if (!tables) tables = img;
  v.nstates = img[1];
  v.natoms = img[2];
  v.npreds = img[3];
  v.nna = img[4];
  v.uses = img[5];
  v.init = (const uint32_t*)(tables + img[6]);
  v.t1 = (const uint32_t*)(tables + img[7]);
  v.t2 = (const uint32_t*)(tables + img[8]);
  v.preds = tables + img[9];
  v.atomsig = (const uint32_t*)(tables + img[10]);
  v.act = (const uint32_t*)(tables + img[11]);
  v.cat = (const uint32_t*)(tables + img[14]);
  v.nskip = (uint32_t)img[16];
  // idle state per category, 8 bits each (idle states have ids < 8)
  v.skippack = ((uint32_t)img[17] & 255u) | (((uint32_t)img[18] & 255u) << 8) | (((uint32_t)img[19] & 255u) << 16) |
               (((uint32_t)img[20] & 255u) << 24);
  v.cand0 = (uint32_t)img[21];
  v.cand1 = (uint32_t)img[22];
  v.cand2 = (uint32_t)img[23];
  v.cand3 = (uint32_t)img[24];
  v.word0 = (uint32_t)img[25];
  v.word1 = (uint32_t)img[26];
  v.word2 = (uint32_t)img[27];
  v.word3 = (uint32_t)img[28];
  v.units = (uint32_t)img[31];
  v.chain = (((uint32_t)img[29] >> 16) & 0xFFFFu) | ((((uint32_t)img[30] >> 16) & 15u) << 16) | ((((uint32_t)img[30] >> 21) & 7u) << 20);
  v.chain |= ((((uint32_t)img[30] >> 24) & 3u) << 24) | ((((uint32_t)img[30] >> 26) & 1u) << 23);
  v.sfx = ((uint32_t)img[30] >> 21) & 7u ? (uint32_t)img[img[15] - 1 - (int)(((uint32_t)img[30] >> 20) & 1u)] : 0u;
  {
    const uint32_t lo1 = (uint32_t)img[29] & 255u, hi1 = ((uint32_t)img[29] >> 8) & 255u;
    const uint32_t lo2 = (uint32_t)img[30] & 255u, hi2 = ((uint32_t)img[30] >> 8) & 255u;
    v.r1lo = (0x80u - lo1) * 0x01010101u;
    v.r1hi = (0x7Fu - hi1) * 0x01010101u;
    v.r2lo = (0x80u - lo2) * 0x01010101u;
    v.r2hi = (0x7Fu - hi2) * 0x01010101u;
  }
  return v;
}

// Executor with the interface the row drivers of regex_vm.h expect
// (find / char_at / s / n).  P is the list-simulator's program view: it is only
// consulted for non-ASCII characters (class membership, word-ness).
struct Tdfa {
  const View& D;
  const csvm::ProgView& P;
  const uint8_t* s;
  int n;
  int sa;  // (address of s) & 3: lets the row be read through aligned 32-bit loads without pointer<->integer casts
  bool wide_ok = true;  // 96 bytes from the row's aligned start are readable (LDS tiles; buffer slack in HBM)

  CS_HD Tdfa(const View& d, const csvm::ProgView& p, const uint8_t* row, int bytes, int align_phase = -1)
      : D(d), P(p), s(row), n(bytes), sa(align_phase < 0 ? (int)((uintptr_t)row & 3) : align_phase) {}

  CS_HD csrow::Char char_at(int i, unsigned& w) const {
    if (i >= n) {
      w = 1;
      return 0;
    }
    csrow::Char c;
    w = csrow::decode_at(s, i, n, c);
    if (w == 0) w = 1;
    // a multi-byte sequence cut off by the end of the row (malformed input) must not carry the
    // scan position past the row: match spans stay inside [0, n], so the size pass and the
    // write pass of every replace kernel agree byte for byte
    if (i + (int)w > n) w = (unsigned)(n - i);
    return c;
  }
  CS_HD unsigned cat_of_ascii(unsigned b) const { return (D.cat[b >> 2] >> (8 * (b & 3))) & 3u; }
  // category index of the character before byte offset i: bit0 word, bit1 newline, bit2 = row start
  CS_HD unsigned prev_cat(int i) const {
    if (i <= 0) return 4u;
    int q = i - 1;
    while (q > 0 && csrow::is_cont(s[q])) --q;
    uint8_t b = s[q];
    if (b < 128) return cat_of_ascii(b);
    csrow::Char c;
    csrow::decode_at(s, q, n, c);
    return csvm::is_word(P, c) ? 1u : 0u;
  }
  CS_HD bool pred_true(int type, int arg, csrow::Char c) const {
    switch (type) {
      case P_CHAR: return c == (csrow::Char)arg;
      case P_ANY: return c != '\n';
      case P_ANYNL: return true;
      case P_CCLASS: return csvm::class_match(P, arg, c);
      case P_NCCLASS: return !csvm::class_match(P, arg, c);
      case P_ISNL: return c == '\n';
      case P_ISWORD: return csvm::is_word(P, c);
    }
    return false;
  }
  CS_HD int nonascii_atom(csrow::Char c) const {
    if (D.nna == 1) return (int)D.atomsig[2];
    uint32_t lo = 0, hi = 0;
    for (int i = 0; i < D.npreds; ++i) {
      if (pred_true(D.preds[2 * i], D.preds[2 * i + 1], c)) {
        if (i < 32) lo |= 1u << i;
        else hi |= 1u << (i - 32);
      }
    }
    for (int k = 0; k < D.nna; ++k)
      if (D.atomsig[3 * k] == lo && D.atomsig[3 * k + 1] == hi) return (int)D.atomsig[3 * k + 2];
    return (int)D.atomsig[2];  // unreachable for a complete atom table
  }

  static CS_HD uint32_t bm128(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, unsigned b) {
    uint32_t lo = (b & 32u) ? w1 : w0, hi = (b & 32u) ? w3 : w2;
    return (((b & 64u) ? hi : lo) >> (b & 31u)) & 1u;
  }
  // Leftmost-first match whose start lies in [from, win_end); win_end is either
  // the row length (search) or from + 1 (anchored), as in the row drivers.
  template <int NS = kMaxSlots>
  CS_HD int find(int from, int win_end, int& mb, int& me) {
    const int mode = (win_end == from + 1) ? MODE_SEED_ONCE : MODE_RESTART;
    uint32_t state = D.init[mode * 8 + (D.uses ? prev_cat(from) : 0u)];
    int st[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) st[j] = from;
    int matched = 0;
    int pos = from;
    // applies transition `e` taken at `pos`; returns true when the automaton stops
    auto apply = [&](uint32_t e) -> bool {
      if (e & E_MATCH) {
        uint32_t o = e_match_origin(e);
        int v = pos;
#pragma unroll
        for (int j = 0; j < NS; ++j)
          if (o == (uint32_t)j) v = st[j];
        mb = v;
        me = pos;
        matched = 1;
      }
      if (e & E_COMPLEX) {
        uint32_t og = D.act[e >> 21];
        int nst[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          uint32_t o = (og >> (4 * j)) & 15u;
          int v = pos;
#pragma unroll
          for (int i = 0; i < NS; ++i)
            if (o == (uint32_t)i) v = st[i];
          nst[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) st[j] = nst[j];
      } else {
        uint32_t keep = e_keep(e);
        if (keep != 15u) {
#pragma unroll
          for (int j = 0; j < NS; ++j)
            if ((uint32_t)j >= keep) st[j] = pos;
        }
      }
      state = e & E_STATE;
      return (e & E_STOP) != 0;
    };
    // one input byte at `pos` (< n); returns true when the automaton stops
    auto feed = [&](uint8_t b) -> bool {
      if (state < D.nskip && b < 128 && !bm128(D.cand0, D.cand1, D.cand2, D.cand3, b)) {
        // idle and this byte cannot start a match: stay idle, no table access
        unsigned cat = 0;
        if (D.uses & 1) cat |= bm128(D.word0, D.word1, D.word2, D.word3, b);
        if (D.uses & 2) cat |= (b == '\n') ? 2u : 0u;
        state = (D.skippack >> (8 * cat)) & 255u;
        ++pos;
        return false;
      }
      int w = 1;
      uint32_t e;
      if (b < 128) {
        e = D.t1[state * 128 + b];
      } else {
        unsigned uw;
        csrow::Char c = char_at(pos, uw);
        w = (int)uw;
        e = D.t2[state * D.natoms + nonascii_atom(c)];
      }
      if (apply(e)) return true;
      pos += w;
      return false;
    };
#if defined(__HIP_DEVICE_COMPILE__)
    // the row is read through aligned 32-bit words; the word is re-read only when
    // the scan crosses into the next one
    const uint32_t* words = reinterpret_cast<const uint32_t*>(s - sa);
    bool stop = false;
    int widx = -1;
    uint32_t word = 0;
    while (!stop && pos < n) {
      const int j = pos + sa;
      if ((j >> 2) != widx) {
        widx = j >> 2;
        word = words[widx];
      }
      stop = feed((uint8_t)(word >> (8 * (j & 3))));
    }
    if (!stop) apply(D.t2[state * D.natoms + ATOM_EOT]);
#else
    bool stop = false;
    while (!stop && pos < n) stop = feed(s[pos]);
    if (!stop) apply(D.t2[state * D.natoms + ATOM_EOT]);
#endif
    return matched;
  }

  // ---- capture groups (regexec.inl:204-442 with groupId != 0; dreprog::extract, regexec.inl:465-469) ----
  // The program run from byte offset `from` only (MODE_SEED_ONCE), every thread slot carrying the
  // (begin, end) range of ONE capture group instead of its start offset.  `G` is the group-tag image
  // of build_tdfa and `group` is 1-based: per (state, atom) a word says which surviving slots -- and the
  // matching thread -- passed that group's LBRA / RBRA in the closure at this position (bits 2j / 2j+1
  // for slot j, bits 8 / 9 for the match), where the simulator sets begin / end to the position.
  // Returns 1 and the range (byte offsets, -1 = never set) of the thread whose END wins, or 0.
  CS_HD int group_find(int from, const int32_t* G, int group, int& gb, int& ge) {
    // rows below 255 bytes keep the four (begin, end) pairs in two registers, a byte per slot (255 = never set)
    return n < 255 ? group_find_impl<true>(from, G, group, gb, ge) : group_find_impl<false>(from, G, group, gb, ge);
  }
  template <bool PACKED>
  CS_HD int group_find_impl(int from, const int32_t* G, int group, int& gb, int& ge) {
    const uint32_t* tags = (const uint32_t*)(G + 36) + (long long)(group - 1) * (long long)G[3];
    const uint32_t* amap = (const uint32_t*)(G + 4);
    uint32_t state = D.init[MODE_SEED_ONCE * 8 + (D.uses ? prev_cat(from) : 0u)];
    int bx[kMaxSlots], by[kMaxSlots];
    uint32_t px = 0xFFFFFFFFu, py = 0xFFFFFFFFu;  // PACKED
#pragma unroll
    for (int j = 0; j < kMaxSlots; ++j) bx[j] = by[j] = -1;
    int matched = 0;
    int pos = from;
    auto pick = [&](const int* v, uint32_t o) -> int {
      int r = -1;
#pragma unroll
      for (int j = 0; j < kMaxSlots; ++j)
        if (o == (uint32_t)j) r = v[j];
      return r;
    };
    auto pick8 = [](uint32_t p, uint32_t o) -> uint32_t { return o > 3u ? 255u : (p >> (8u * o)) & 255u; };
    auto apply = [&](uint32_t e, uint32_t tg) -> bool {
      // most transitions inside a match keep every slot where it is and pass no bracket of the group
      if (tg == 0 && (e & (E_MATCH | E_COMPLEX | (15u << 16))) == 0) {
        state = e & E_STATE;
        return (e & E_STOP) != 0;
      }
      if (e & E_MATCH) {
        const uint32_t o = e_match_origin(e);
        if (PACKED) {
          const uint32_t mx = (tg & 0x100u) ? (uint32_t)pos : pick8(px, o), my = (tg & 0x200u) ? (uint32_t)pos : pick8(py, o);
          gb = mx == 255u ? -1 : (int)mx;
          ge = my == 255u ? -1 : (int)my;
        } else {
          const int mx = pick(bx, o), my = pick(by, o);
          gb = (tg & 0x100u) ? pos : mx;
          ge = (tg & 0x200u) ? pos : my;
        }
        matched = 1;
      }
      uint32_t og = 0x3210u;  // identity
      if (e & E_COMPLEX) {
        og = D.act[e >> 21];
      } else {
        const uint32_t keep = e_keep(e);
        if (keep != 15u) og = (0x3210u & ~(0xFFFFu << (4 * keep))) | (0xFFFFu << (4 * keep));
      }
      if (PACKED) {
        uint32_t nx = 0, ny = 0;
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j) {
          const uint32_t o = (og >> (4 * j)) & 15u;
          nx |= (((tg >> (2 * j)) & 1u) ? (uint32_t)pos : pick8(px, o)) << (8 * j);
          ny |= (((tg >> (2 * j + 1)) & 1u) ? (uint32_t)pos : pick8(py, o)) << (8 * j);
        }
        px = nx;
        py = ny;
      } else {
        int nx[kMaxSlots], ny[kMaxSlots];
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j) {
          const uint32_t o = (og >> (4 * j)) & 15u;
          nx[j] = ((tg >> (2 * j)) & 1u) ? pos : pick(bx, o);
          ny[j] = ((tg >> (2 * j + 1)) & 1u) ? pos : pick(by, o);
        }
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j) {
          bx[j] = nx[j];
          by[j] = ny[j];
        }
      }
      state = e & E_STATE;
      return (e & E_STOP) != 0;
    };
    bool stop = false;
    while (!stop && pos < n) {
      const uint8_t b = byte_at(pos);
      int w = 1;
      uint32_t e, atom;
      if (b < 128) {
        e = D.t1[state * 128 + b];
        atom = (amap[b >> 2] >> (8 * (b & 3))) & 255u;
      } else {
        unsigned uw;
        const csrow::Char c = char_at(pos, uw);
        w = (int)uw;
        atom = (uint32_t)nonascii_atom(c);
        e = D.t2[state * D.natoms + atom];
      }
      stop = apply(e, tags[state * D.natoms + atom]);
      if (!stop) pos += w;
    }
    if (!stop) apply(D.t2[state * D.natoms + ATOM_EOT], tags[state * D.natoms + ATOM_EOT]);
    return matched;
  }

  // All capture groups of a match from ONE anchored run: groups first .. first + count - 1 (count <= kGroupBatch),
  // ranges into gb[] / ge[] (-1 = never set), `mend` = where the match ends (the range of group 0).  Same
  // automaton, same slot shuffles as group_find; every group keeps its own (begin, end) per slot, a byte each,
  // so rows of 255 bytes and more take group_find per group (the callers' choice).  The fast path -- a
  // transition that moves no slot and passes no bracket of ANY tracked group -- costs one more table word per
  // group per step; everything else is paid once instead of once per group.
  static constexpr int kGroupBatch = 4;
  // v_perm_b32: result byte j = byte sel[j] of {hi : lo} for sel[j] in 0..7, 0x00 for 12, 0xFF for 13 and above
  // (the selector values 8..11 are not used here)
  static CS_HD uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint32_t r = 0;
    for (int j = 0; j < 4; ++j) {
      const uint32_t c = (sel >> (8 * j)) & 255u;
      const uint32_t b = c < 4 ? (lo >> (8 * c)) & 255u : c < 8 ? (hi >> (8 * (c - 4))) & 255u : c == 12 ? 0u : 255u;
      r |= b << (8 * j);
    }
    return r;
#endif
  }
  CS_HD int group_find_all(int from, const int32_t* G, int first, int count, int* gb, int* ge, int& mend) {
    const uint32_t* tags0 = (const uint32_t*)(G + 36) + (long long)(first - 1) * (long long)G[3];
    const long long tstride = (long long)G[3];
    const uint32_t* amap = (const uint32_t*)(G + 4);
    uint32_t state = D.init[MODE_SEED_ONCE * 8 + (D.uses ? prev_cat(from) : 0u)];
    uint32_t px[kGroupBatch], py[kGroupBatch];
#pragma unroll
    for (int g = 0; g < kGroupBatch; ++g) {
      px[g] = py[g] = 0xFFFFFFFFu;
      gb[g] = ge[g] = -1;
    }
    int matched = 0;
    int pos = from;
    mend = from;
    auto pick8 = [](uint32_t p, uint32_t o) -> uint32_t { return o > 3u ? 255u : (p >> (8u * o)) & 255u; };
    auto apply = [&](uint32_t e, const uint32_t* tg, uint32_t any) -> bool {
      if (any == 0 && (e & (E_MATCH | E_COMPLEX | (15u << 16))) == 0) {
        state = e & E_STATE;
        return (e & E_STOP) != 0;
      }
      if (e & E_MATCH) {
        const uint32_t o = e_match_origin(e);
#pragma unroll
        for (int g = 0; g < kGroupBatch; ++g)
          if (g < count) {
            const uint32_t mx = (tg[g] & 0x100u) ? (uint32_t)pos : pick8(px[g], o), my = (tg[g] & 0x200u) ? (uint32_t)pos : pick8(py[g], o);
            gb[g] = mx == 255u ? -1 : (int)mx;
            ge[g] = my == 255u ? -1 : (int)my;
          }
        mend = pos;
        matched = 1;
      }
      if (!(e & E_COMPLEX) && e_keep(e) == 15u) {
        // every slot stays where it is: only the tagged slots take the position (a byte mask per tag nibble)
        const uint32_t posb = (uint32_t)pos * 0x01010101u;
        auto spread = [](uint32_t x) -> uint32_t {  // bits 0, 2, 4, 6 -> bytes 0..3 all ones
          return (((x | (x << 6) | (x << 12) | (x << 18)) & 0x01010101u)) * 255u;
        };
#pragma unroll
        for (int g = 0; g < kGroupBatch; ++g)
          if (g < count && (tg[g] & 0xFFu)) {
            const uint32_t bm = spread(tg[g] & 0x55u), em = spread((tg[g] >> 1) & 0x55u);
            px[g] = (px[g] & ~bm) | (posb & bm);
            py[g] = (py[g] & ~em) | (posb & em);
          }
        state = e & E_STATE;
        return (e & E_STOP) != 0;
      }
      uint32_t og = 0x3210u;  // identity
      if (e & E_COMPLEX) {
        og = D.act[e >> 21];
      } else {
        const uint32_t keep = e_keep(e);
        if (keep != 15u) og = (0x3210u & ~(0xFFFFu << (4 * keep))) | (0xFFFFu << (4 * keep));
      }
      // The slots move as whole bytes: ONE byte permute per group and side (v_perm_b32).  Selector byte j = the
      // origin slot (0..3: that byte of the old word), 15 for a new thread (-> 0xFF, never set), 4 where the
      // slot passes the group's bracket here (-> a byte of `posb`, the position).
      const uint32_t sel = (og & 0xFu) | ((og & 0xF0u) << 4) | ((og & 0xF00u) << 8) | ((og & 0xF000u) << 12);
      const uint32_t posb = (uint32_t)pos * 0x01010101u;
      auto spread = [](uint32_t x) -> uint32_t {  // bits 0, 2, 4, 6 -> bytes 0..3 all ones
        return (((x | (x << 6) | (x << 12) | (x << 18)) & 0x01010101u)) * 255u;
      };
#pragma unroll
      for (int g = 0; g < kGroupBatch; ++g)
        if (g < count) {
          const uint32_t bm = spread(tg[g] & 0x55u), em = spread((tg[g] >> 1) & 0x55u);
          px[g] = perm_bytes(posb, px[g], (sel & ~bm) | (0x04040404u & bm));
          py[g] = perm_bytes(posb, py[g], (sel & ~em) | (0x04040404u & em));
        }
      state = e & E_STATE;
      return (e & E_STOP) != 0;
    };
    auto step = [&](uint32_t e, uint32_t atom) -> bool {
      uint32_t tg[kGroupBatch], any = 0;
      const uint32_t* t = tags0 + state * D.natoms + atom;
#pragma unroll
      for (int g = 0; g < kGroupBatch; ++g) {
        tg[g] = g < count ? t[g * tstride] : 0u;
        any |= tg[g];
      }
      return apply(e, tg, any);
    };
    bool stop = false;
    while (!stop && pos < n) {
      const uint8_t b = byte_at(pos);
      int w = 1;
      uint32_t e, atom;
      if (b < 128) {
        e = D.t1[state * 128 + b];
        atom = (amap[b >> 2] >> (8 * (b & 3))) & 255u;
      } else {
        unsigned uw;
        const csrow::Char c = char_at(pos, uw);
        w = (int)uw;
        atom = (uint32_t)nonascii_atom(c);
        e = D.t2[state * D.natoms + atom];
      }
      stop = step(e, atom);
      if (!stop) pos += w;
    }
    if (!stop) step(D.t2[state * D.natoms + ATOM_EOT], (uint32_t)ATOM_EOT);
    return matched;
  }

  // The same result by a BACKWARD walk (matches of up to kBackSteps ASCII bytes on automata of up to 256 states;
  // -1 = not applicable, the caller takes group_find_all).  The forward run above carries the (begin, end) of every
  // group for every live thread through every step -- a byte permute per group and side per step.  Only ONE thread
  // matters in the end, the one that matches: here the forward run is the plain automaton (one table word per byte,
  // the state before each step kept in `hist`, a lane-private byte array), and the groups are read off on the way
  // back from the match along that thread's origin slots: at step i the slot `cur` either passed a bracket (its
  // two tag bits per group, all four groups of the batch in one word of the packed table) -- then the position of
  // step i is that bracket's final value unless a later step already set it -- or it inherits from slot
  // origin_i[cur] of the step before.
  static constexpr int kBackSteps = 32;
  struct HistBytes {  // `hist` over plain memory (LDS on the device: a lane's bytes lie `stride` apart)
    uint8_t* p;
    int stride;
    CS_HD void put(int i, uint8_t v) { p[i * stride] = v; }
    CS_HD uint32_t get(int i) const { return p[i * stride]; }
  };
  template <class Hist>
  CS_HD int group_find_back(int from, const int32_t* G, int first, int count, int* gb, int* ge, int& mend, Hist hist, int max_steps = kBackSteps) {
    if (D.nstates > 256) return -1;
    // the packed tag table holds a BATCH of four groups per word at batch-relative bit positions: only whole batches
    // (groups 1-4, 5-8, ...) can be resolved this way -- anything else goes to the forward run
    if ((first - 1) % kGroupBatch != 0 || count > kGroupBatch || count < 0) return -1;
    const long long tstride = (long long)G[3];
    const uint32_t* tags0 = (const uint32_t*)(G + 36) + (long long)(first - 1) * tstride;
    const uint32_t* packed = (const uint32_t*)(G + 36) + ((long long)G[0] + (first - 1) / kGroupBatch) * tstride;
    const uint32_t* amap = (const uint32_t*)(G + 4);
    uint32_t state = D.init[MODE_SEED_ONCE * 8 + (D.uses ? prev_cat(from) : 0u)];
#pragma unroll
    for (int g = 0; g < kGroupBatch; ++g) gb[g] = ge[g] = -1;
    mend = from;
    int steps = 0, m_step = -1;
    uint32_t m_origin = 15u, m_state = 0, m_atom = 0;
    bool stop = false;
    int pos = from;
    while (!stop && pos < n) {
      const uint8_t b = byte_at(pos);
      if (b >= 128 || steps >= max_steps) return -1;  // (`max_steps`: what the caller's history holds)
      const uint32_t e = D.t1[state * 128 + b];
      if (e & E_MATCH) {
        m_step = steps;
        m_origin = e_match_origin(e);
        m_state = state;
        m_atom = (amap[b >> 2] >> (8 * (b & 3))) & 255u;
        mend = pos;
      }
      hist.put(steps, (uint8_t)state);
      ++steps;
      state = e & E_STATE;
      stop = (e & E_STOP) != 0;
      if (!stop) ++pos;
    }
    if (!stop) {
      const uint32_t e = D.t2[state * D.natoms + ATOM_EOT];
      if (e & E_MATCH) {
        m_step = steps;
        m_origin = e_match_origin(e);
        m_state = state;
        m_atom = (uint32_t)ATOM_EOT;
        mend = pos;
      }
    }
    if (m_step < 0) return 0;
    uint32_t open = 0;  // bit 2q / 2q + 1: begin / end of the batch's q-th group not set yet
#pragma unroll
    for (int g = 0; g < kGroupBatch; ++g)
      if (g < count) {
        const uint32_t tg = tags0[g * tstride + m_state * D.natoms + m_atom];
        if (tg & 0x100u) gb[g] = mend;
        else open |= 1u << (2 * g);
        if (tg & 0x200u) ge[g] = mend;
        else open |= 2u << (2 * g);
      }
    uint32_t cur = m_origin;
    for (int i = m_step - 1; i >= 0 && cur < 4u && open; --i) {
      const uint32_t s = hist.get(i);
      const uint8_t b = byte_at(from + i);
      const uint32_t e = D.t1[s * 128 + b];
      const uint32_t atom = (amap[b >> 2] >> (8 * (b & 3))) & 255u;
      const uint32_t hit = (packed[s * D.natoms + atom] >> (8 * cur)) & open;
      if (hit) {
#pragma unroll
        for (int g = 0; g < kGroupBatch; ++g) {
          if (hit & (1u << (2 * g))) gb[g] = from + i;
          if (hit & (2u << (2 * g))) ge[g] = from + i;
        }
        open &= ~hit;
      }
      if (e & E_COMPLEX) {
        cur = (D.act[e >> 21] >> (4 * cur)) & 15u;
      } else {
        const uint32_t keep = e_keep(e);
        if (keep != 15u && cur >= keep) cur = 15u;
      }
    }
    return 1;
  }

  // ---- flat scan: all successive matches of a row in ONE loop ------------------
  // (the SIMT-friendly form of the row drivers in regex_vm.h: lanes that are in
  // different find() rounds still share the loop body, and idle stretches are
  // skipped a 32-bit word at a time)
  enum { K_CONTAINS = 0, K_MATCH = 1, K_COUNT = 2, K_REPLACE = 3 };

  // aligned 32-bit word `widx` of the row's storage; bytes outside the row may be anything
  CS_HD uint32_t load_word(int widx) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return reinterpret_cast<const uint32_t*>(s - sa)[widx];
#else
    uint32_t w = 0;
    for (int k = 0; k < 4; ++k) {
      int i = widx * 4 + k - sa;
      w |= (uint32_t)((i >= 0 && i < n) ? s[i] : 0xFFu) << (8 * k);
    }
    return w;
#endif
  }
  // bit 7 of each byte lane set when that byte may leave the idle state
  CS_HD uint32_t cand_mask(uint32_t w) const {
    const uint32_t x = w & 0x7F7F7F7Fu;
    uint32_t m = w | ((w - 0x01010101u) & ~w);                 // non-ASCII, NUL (conservative)
    m |= (x + D.r1lo) & ~(x + D.r1hi);
    m |= (x + D.r2lo) & ~(x + D.r2hi);
    return m & 0x80808080u;
  }
  static CS_HD int ctz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ctz(v);
#else
    int k = 0;
    while (!(v & 1u)) {
      v >>= 1;
      ++k;
    }
    return k;
#endif
  }

  // ---- candidate bitmask of a short row, held in three registers ---------------
  // Bit q = "the byte at row offset q - sa may leave the idle state", q < 96.  Built
  // in straight-line code from 24 aligned words (no loop, no waits between the
  // loads), after which "next candidate" is a couple of bit operations.
  static constexpr int kMaskBytes = 96;
  CS_HD bool masks_fit() const { return wide_ok && n + sa <= kMaskBytes; }
  CS_HD void build_masks(uint32_t& m0, uint32_t& m1, uint32_t& m2) const {
    uint32_t r[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const uint32_t c = cand_mask(load_word(k));
      // gather bits 7,15,23,31 into a nibble (bit i = byte i)
      const uint32_t nib = ((((c >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u;
      r[k >> 3] |= nib << (4 * (k & 7));
    }
    // keep bits sa .. sa + n - 1
    const int lo = sa, hi = sa + n;
    r[0] &= 0xFFFFFFFFu << lo;
    r[0] &= hi >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (hi & 31));
    r[1] &= hi >= 64 ? 0xFFFFFFFFu : (hi <= 32 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
    r[2] &= hi >= 96 ? 0xFFFFFFFFu : (hi <= 64 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
    m0 = r[0];
    m1 = r[1];
    m2 = r[2];
  }
  // Row-relative candidate bits for the lean scan: the caller guarantees that every byte the
  // 24 words hold inside the sub-tile is 1..127 (no non-ASCII / NUL terms needed) and
  // HAS_R2 says whether the second byte range is in use.
  template <bool HAS_R2>
  CS_HD void build_masks_lean(uint32_t& m0, uint32_t& m1, uint32_t& m2) const {
    uint32_t r[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const uint32_t x = load_word(k) & 0x7F7F7F7Fu;
      uint32_t c = (x + D.r1lo) & ~(x + D.r1hi);
      if (HAS_R2) c |= (x + D.r2lo) & ~(x + D.r2hi);
      c &= 0x80808080u;
      const uint32_t nib = ((((c >> 7)) * 0x01020408u) >> 24) & 15u;  // bits 7,15,23,31 -> nibble
      r[k >> 3] |= nib << (4 * (k & 7));
    }
    // shift the aligned-start masks down to the row's first byte and cut at its length
    const unsigned a = (unsigned)sa;
    uint32_t q0 = a ? (r[0] >> a) | (r[1] << (32 - a)) : r[0];
    uint32_t q1 = a ? (r[1] >> a) | (r[2] << (32 - a)) : r[1];
    uint32_t q2 = r[2] >> a;
    const int hi = n;
    q0 &= hi >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (hi & 31));
    q1 &= hi >= 64 ? 0xFFFFFFFFu : (hi <= 32 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
    q2 &= hi >= 96 ? 0xFFFFFFFFu : (hi <= 64 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
    m0 = q0;
    m1 = q1;
    m2 = q2;
  }
  // bit 7 of each byte lane: the (ASCII, non-NUL) byte is a candidate
  template <bool HAS_R2>
  static CS_HD uint32_t cand_bits_ascii(const View& V, uint32_t w) {
    const uint32_t x = w & 0x7F7F7F7Fu;
    uint32_t c = (x + V.r1lo) & ~(x + V.r1hi);
    if (HAS_R2) c |= (x + V.r2lo) & ~(x + V.r2hi);
    return c & 0x80808080u;
  }
  CS_HD bool has_range2() const { return ((uint32_t)D.img[30] & 255u) <= (((uint32_t)D.img[30] >> 8) & 255u); }
  // first candidate position >= pos (row offsets), or n
  CS_HD int next_candidate(uint32_t m0, uint32_t m1, uint32_t m2, int pos) const {
    const int q = pos + sa;
    uint32_t a = m0, b = m1, c = m2;
    if (q >= 32) a = 0;
    if (q >= 64) b = 0;
    const uint32_t cut = 0xFFFFFFFFu << (q & 31);
    if (q < 32) a &= cut;
    else if (q < 64) b &= cut;
    else c &= cut;
    if (a) return ctz32(a) - sa;
    if (b) return 32 + ctz32(b) - sa;
    if (c) return 64 + ctz32(c) - sa;
    return n;
  }


  // ---- lean replace scan: ASCII-only rows, candidate masks in registers ------------
  // Preconditions (the caller checks them): every byte of the row is 1..127,
  // D.nskip > 0, n <= kMaskBytes.  (m0, m1, m2) hold the row's candidate bits, bit i =
  // "byte i may leave the idle state".  On such rows the result equals scan<K_REPLACE>.
  // Two levels so that a wave spends its lock-step iterations on byte steps, not on
  // bookkeeping: the inner loop takes every transition that neither stops the automaton
  // nor returns it to the idle state, applying MATCH / thread-start side effects without
  // branches; idle jumps, the end of the row and the end of a find() round are handled
  // once per outer iteration.  The (up to four) thread start offsets live in ONE register,
  // a byte each (row offsets are below 256): "slots j >= keep start here" is a single
  // bit-field insert and a match origin a single bit-field extract.
  // COMPLEX transitions and zero-length matches set `bail`: the caller re-runs the row
  // with scan<>.
  CS_HD uint8_t byte_at(int i) const { return s[i]; }
  static CS_HD int first_candidate(uint32_t m0, uint32_t m1, uint32_t m2, int pos, int n) {
    uint32_t a = m0, b = m1, c = m2;
    if (pos >= 32) a = 0;
    if (pos >= 64) b = 0;
    const uint32_t cut = 0xFFFFFFFFu << (pos & 31);
    if (pos < 32) a &= cut;
    else if (pos < 64) b &= cut;
    else c &= cut;
    if (a) return ctz32(a);
    if (b) return 32 + ctz32(b);
    if (c) return 64 + ctz32(c);
    return n;
  }
  int lean_resume_from = 0;  // set with `bail`: the find() round that has to be redone starts here
  struct NoRefill {
    CS_HD void operator()(int, uint32_t&, uint32_t&, uint32_t&) const {}
  };
  // LONG: rows of up to kLongBytes bytes.  (cm0, cm1, cm2) are then a 96-byte WINDOW of the
  // row's candidate bits starting at row offset `wb`; refill(wb, m0, m1, m2) fetches the window
  // at a new base (the kernels keep the whole tile's bits in an LDS bitmap).
  static constexpr int kLongBytes = 255;  // thread start offsets are kept one byte each
  template <int KIND, bool USES, class Emit, bool LONG = false, class Refill = NoRefill>
  CS_HD int scan_lean(int maxrepl, uint32_t cm0, uint32_t cm1, uint32_t cm2, Emit&& emit, bool& bail, Refill refill = Refill()) {
    int wb = 0;  // LONG: row offset of bit 0 of the candidate window
    int from = 0, pos = 0, done = 0, mb = 0, me = 0;
    uint32_t matched = 0;
    uint32_t slots = 0;  // byte j = start offset of thread slot j
    uint32_t state = D.init[MODE_RESTART * 8 + 4];
    // side effects of a transition without a COMPLEX origin word, branch-free
    uint32_t posb = 0;  // pos in every byte lane (kept in step with pos: no multiply in the inner loop)
    auto effects = [&](uint32_t e) {
      const uint32_t is_m = (e >> 11) & 1u;
      const uint32_t o = (e >> 12) & 15u;
      const uint32_t from_slot = (slots >> (8u * (o & 3u))) & 255u;
      const int v = (o & 8u) ? pos : (int)from_slot;  // origin 15 = thread born here
      mb = is_m ? v : mb;
      me = is_m ? pos : me;
      matched |= is_m;
      const uint32_t kf = (e >> 16) & 15u;  // keep ^ 15: 0 = nothing to do
      const uint32_t keep = kf ^ 15u;
      const uint32_t low = keep >= 4u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8u * keep));  // bytes below `keep` stay
      slots = (slots & low) | (posb & ~low);
    };
    for (;;) {
      if (state < D.nskip && pos < n) {  // idle: jump to the next candidate byte
        const int entry = pos;
        if (LONG) {
          for (;;) {
            if (pos < wb || pos - wb >= 96) {  // the window does not cover pos (a new round may step back)
              wb = pos;
              refill(wb, cm0, cm1, cm2);
            }
            const int lim = n - wb < 96 ? n - wb : 96;
            const int c = first_candidate(cm0, cm1, cm2, pos - wb, lim);
            pos = wb + (c < lim ? c : lim);
            if (c < lim || pos >= n) break;
          }
        } else {
          pos = first_candidate(cm0, cm1, cm2, pos, n);
        }
        if (pos > n) pos = n;
        posb = (uint32_t)pos * 0x01010101u;
        if (USES && pos > entry) {
          const unsigned b = byte_at(pos - 1);
          unsigned cat = 0;
          if (D.uses & 1) cat |= bm128(D.word0, D.word1, D.word2, D.word3, b);
          if (D.uses & 2) cat |= (b == '\n') ? 2u : 0u;
          state = (D.skippack >> (8 * cat)) & 255u;
        }
      }
      uint32_t e;
      if (pos >= n) {
        e = D.t2[state * D.natoms + ATOM_EOT];
      } else {
        unsigned b = byte_at(pos);
        for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
          const unsigned bn = byte_at(pos + 1);  // next byte in flight together with the table lookup
#endif
          e = D.t1[state * 128 + b];
          if ((e & (E_STOP | E_COMPLEX | E_EXIT)) || pos + 1 >= n) break;
          effects(e);
          state = e & E_STATE;
          ++pos;
          posb += 0x01010101u;
#if defined(__HIP_DEVICE_COMPILE__)
          b = bn;
#else
          b = byte_at(pos);
#endif
        }
      }
      if (e & E_COMPLEX) {
        bail = true;
        lean_resume_from = from;
        return done;
      }
      effects(e);
      if (!(e & E_STOP)) {
        state = e & E_STATE;
        pos += 1;
        posb += 0x01010101u;
        continue;
      }
      // ---- this find() round is over
      if (!matched) return done;
      if (KIND == K_CONTAINS) {
        emit(mb, me, 1);  // (extract: the leftmost match; contains_re passes a no-op)
        return 1;
      }
      if (KIND == K_COUNT) {
        emit(mb, me, 1);  // (findall; count_re passes a no-op)
        ++done;
        from = me > mb ? me : mb + 1;  // empty match: step one (ASCII) character (count.cu:190-196)
        if (from > n) return done;
      } else {
        if (me == mb && mb == from) {  // zero-length repeat rule (replace.cu:91-93): generic path
          bail = true;
          lean_resume_from = from;
          return done;
        }
        emit(mb, me, 1);
        ++done;
        if (maxrepl >= 0 && done >= maxrepl) return done;
        from = me;
      }
      pos = from;
      matched = 0;
      posb = (uint32_t)from * 0x01010101u;
      slots = posb;
      unsigned pc = 0;
      if (USES) pc = from <= 0 ? 4u : cat_of_ascii(byte_at(from - 1));
      state = D.init[MODE_RESTART * 8 + pc];
    }
  }
  // true when the row can take scan_lean_replace (host-side check; kernels decide per tile)
  CS_HD bool lean_ok() const {
    if (D.nskip == 0 || !masks_fit() || D.img[12] > 4) return false;
    for (int i = 0; i < n; ++i)
      if (s[i] == 0 || s[i] >= 128) return false;
    return true;
  }
  template <class Emit>
  CS_HD int scan_lean_dispatch(int maxrepl, uint32_t m0, uint32_t m1, uint32_t m2, Emit&& emit, bool& bail) {
    if (D.uses) return scan_lean<K_REPLACE, true>(maxrepl, m0, m1, m2, emit, bail);
    return scan_lean<K_REPLACE, false>(maxrepl, m0, m1, m2, emit, bail);
  }
  template <class Emit, class Refill>
  CS_HD int scan_lean_dispatch_long(int maxrepl, uint32_t m0, uint32_t m1, uint32_t m2, Emit&& emit, bool& bail, Refill refill) {
    if (D.uses) return scan_lean<K_REPLACE, true, Emit&, true, Refill>(maxrepl, m0, m1, m2, emit, bail, refill);
    return scan_lean<K_REPLACE, false, Emit&, true, Refill>(maxrepl, m0, m1, m2, emit, bail, refill);
  }
  template <int KIND, class Refill>
  CS_HD int scan_lean_count_long(uint32_t m0, uint32_t m1, uint32_t m2, Refill refill) {
    bool bail = false;
    auto none = [](int, int, int) {};
    const int r = D.uses ? scan_lean<KIND, true, decltype(none)&, true, Refill>(0, m0, m1, m2, none, bail, refill)
                         : scan_lean<KIND, false, decltype(none)&, true, Refill>(0, m0, m1, m2, none, bail, refill);
    return bail ? -1 : r;
  }
  // extract: the leftmost match through emit(mb, me, 1); 1 / 0 = match / none, -1 = the row needs the generic scan
  template <class Emit>
  CS_HD int scan_lean_first(uint32_t m0, uint32_t m1, uint32_t m2, Emit&& emit) {
    bool bail = false;
    const int r = D.uses ? scan_lean<K_CONTAINS, true>(0, m0, m1, m2, emit, bail) : scan_lean<K_CONTAINS, false>(0, m0, m1, m2, emit, bail);
    return bail ? -1 : r;
  }
  // findall: the count_re walk with emit(mb, me, 1) per match; -1 = the row needs the generic scan
  template <class Emit>
  CS_HD int scan_lean_spans(uint32_t m0, uint32_t m1, uint32_t m2, Emit&& emit) {
    bool bail = false;
    const int r = D.uses ? scan_lean<K_COUNT, true>(0, m0, m1, m2, emit, bail) : scan_lean<K_COUNT, false>(0, m0, m1, m2, emit, bail);
    return bail ? -1 : r;
  }
  template <class Emit, class Refill>
  CS_HD int scan_lean_spans_long(uint32_t m0, uint32_t m1, uint32_t m2, Emit&& emit, Refill refill) {
    bool bail = false;
    const int r = D.uses ? scan_lean<K_COUNT, true, Emit&, true, Refill>(0, m0, m1, m2, emit, bail, refill)
                         : scan_lean<K_COUNT, false, Emit&, true, Refill>(0, m0, m1, m2, emit, bail, refill);
    return bail ? -1 : r;
  }
  // contains_re (KIND = K_CONTAINS) / count_re (K_COUNT) on a qualifying row; -1 = the row needs
  // the generic scan (a COMPLEX transition was met)
  template <int KIND>
  CS_HD int scan_lean_count(uint32_t m0, uint32_t m1, uint32_t m2) {
    bool bail = false;
    auto none = [](int, int, int) {};
    const int r = D.uses ? scan_lean<KIND, true>(0, m0, m1, m2, none, bail) : scan_lean<KIND, false>(0, m0, m1, m2, none, bail);
    return bail ? -1 : r;
  }

  // emit(mb, me, reps) per match (K_REPLACE: reps > 1 for the zero-length repeat);
  // returns the number of matches (K_CONTAINS / K_MATCH: 0 or 1).
  template <int KIND, class Emit, int NS = kMaxSlots>
  CS_HD int scan(int maxrepl, Emit&& emit, int from0 = 0, int done0 = 0) {
    // (from0, done0): resume a K_REPLACE / K_COUNT scan at the start of a find() round -- the
    // state every round starts in depends only on the position (used when the lean scan hands a
    // row over in the middle)
    int from = from0, pos = from0, done = done0;
    int mb = 0, me = 0, matched = 0;
    int st[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) st[j] = from0;
    uint32_t state = from0 == 0 ? D.init[(KIND == K_MATCH ? MODE_SEED_ONCE : MODE_RESTART) * 8 + 4]  // row start
                                : D.init[MODE_RESTART * 8 + (D.uses ? prev_cat(from0) : 0u)];
    // one-word cache of the row storage: consecutive positions share a word
    int cwi = -(1 << 30);
    uint32_t cw = 0;
    auto word_at = [&](int widx) -> uint32_t {
      if (widx != cwi) {
        cwi = widx;
        cw = load_word(widx);
      }
      return cw;
    };
    const bool masked = D.nskip > 0 && masks_fit();
    uint32_t cm0 = 0, cm1 = 0, cm2 = 0;
    if (masked) build_masks(cm0, cm1, cm2);
    for (;;) {
      // ---- idle: jump to the next candidate byte
      if (state < D.nskip && pos < n) {
        const int entry = pos;
        if (masked) {
          pos = next_candidate(cm0, cm1, cm2, pos);
        } else {
          do {
            const int j = pos + sa;
            const uint32_t m = cand_mask(word_at(j >> 2)) & (0xFFFFFFFFu << (8 * (j & 3)));
            if (m) {
              pos = (j & ~3) - sa + (ctz32(m) >> 3);
              break;
            }
            pos = (j & ~3) + 4 - sa;
          } while (pos < n);
        }
        if (pos > n) pos = n;
        if (D.uses && pos > entry) {
          const int j = pos - 1 + sa;
          const unsigned b = (word_at(j >> 2) >> (8 * (j & 3))) & 255u;
          unsigned cat = 0;
          if (b < 128) {
            if (D.uses & 1) cat |= bm128(D.word0, D.word1, D.word2, D.word3, b);
            if (D.uses & 2) cat |= (b == '\n') ? 2u : 0u;
          } else {
            cat = prev_cat(pos) & 3u;  // multi-byte character before pos
          }
          state = (D.skippack >> (8 * cat)) & 255u;
        }
      }
      // ---- one transition at pos; transitions without side effects are taken in a
      // tight inner loop (an entry below 1024 is just the next state)
      uint32_t e;
      int w = 1;
      if (pos >= n) {
        e = D.t2[state * D.natoms + ATOM_EOT];
      } else {
        int j = pos + sa;
        unsigned b = (word_at(j >> 2) >> (8 * (j & 3))) & 255u;
        if (b < 128) {
          e = D.t1[state * 128 + b];
          while (e < 1024u && e >= D.nskip && pos + 1 < n) {
            state = e;
            ++pos;
            ++j;
            b = (word_at(j >> 2) >> (8 * (j & 3))) & 255u;
            if (b >= 128) break;
            e = D.t1[state * 128 + b];
          }
        }
        if (b < 128) {
          // e is the pending transition of (state, byte at pos)
        } else {
          unsigned uw;
          csrow::Char c = char_at(pos, uw);
          w = (int)uw;
          e = D.t2[state * D.natoms + nonascii_atom(c)];
        }
      }
      if (e & E_MATCH) {
        const uint32_t o = e_match_origin(e);
        int v = pos;
#pragma unroll
        for (int j = 0; j < NS; ++j)
          if (o == (uint32_t)j) v = st[j];
        mb = v;
        me = pos;
        matched = 1;
      }
      if (e & E_COMPLEX) {
        const uint32_t og = D.act[e >> 21];
        int nst[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const uint32_t o = (og >> (4 * j)) & 15u;
          int v = pos;
#pragma unroll
          for (int i = 0; i < NS; ++i)
            if (o == (uint32_t)i) v = st[i];
          nst[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) st[j] = nst[j];
      } else {
        const uint32_t keep = e_keep(e);
        if (keep != 15u) {
#pragma unroll
          for (int j = 0; j < NS; ++j)
            if ((uint32_t)j >= keep) st[j] = pos;
        }
      }
      if (!(e & E_STOP)) {
        state = e & E_STATE;
        pos += w;
        continue;
      }
      // ---- this find() round is over
      if (!matched) return done;
      if (KIND == K_CONTAINS || KIND == K_MATCH) return 1;
      if (KIND == K_COUNT) {
        emit(mb, me, 1);  // (findall; count_re passes a no-op)
        ++done;
        if (me > mb) {
          from = me;
        } else {  // empty match: step one character (count.cu:190-196)
          unsigned uw;
          char_at(mb, uw);
          from = mb + (int)uw;
        }
        if (from > n) return done;
      } else {
        if (me == mb && mb == from) {
          // a zero-length match does not advance the search (replace.cu:91-93):
          // the same match repeats until the budget is spent
          const int left = (maxrepl < 0 ? csrow::count_chars(s, n) : maxrepl) - done;
          if (left > 0) emit(mb, me, left);
          return done + (left > 0 ? left : 0);
        }
        emit(mb, me, 1);
        ++done;
        if (maxrepl >= 0 && done >= maxrepl) return done;
        from = me;
      }
      // next round starts at `from`
      pos = from;
      matched = 0;
#pragma unroll
      for (int j = 0; j < NS; ++j) st[j] = from;
      state = D.init[MODE_RESTART * 8 + (D.uses ? prev_cat(from) : 0u)];
    }
  }
};

// The same executor for programs of five to eight live threads (header word 12): only the generic scan and find, with
// eight start offsets.  A distinct type so that the row drivers below pick the right width by overload.
struct TdfaWide : Tdfa {
  using Tdfa::Tdfa;
};

}  // namespace cstd

// Row drivers for the tagged DFA: same contracts as the templates in regex_vm.h
// (overloads, so call sites are engine-agnostic).
namespace csvm {
#if !defined(__HIP_DEVICE_COMPILE__)
// host builds: a plain-ASCII row's matches by chain_match (the route of the stream kernels), or false
inline bool row_chain_host(cstd::Tdfa& vm, cstd::U128& S, cstd::U128& L) {
  if (!(vm.D.chain >> 16) || !cstd::g_chain_host || !vm.lean_ok()) return false;
  uint32_t c0, c1, c2;
  if (vm.has_range2()) vm.build_masks_lean<true>(c0, c1, c2);
  else vm.build_masks_lean<false>(c0, c1, c2);
  const unsigned x = (vm.D.units >> 8) & 127u;
  cstd::U128 X = cstd::u128(0, 0);
  for (int i = 0; x && i < vm.n; ++i)
    if (vm.s[i] == x) {
      if (i < 64) X.lo |= 1ull << i;
      else X.hi |= 1ull << (i - 64);
    }
  if (vm.n <= 95 && cstd::g_chain_host == 1)  // (the three-word form of the replace kernel; g_chain_host 2: the two-half form on every row)
    cstd::chain_match96(c0, c1, c2, (uint32_t)X.lo, (uint32_t)(X.lo >> 32), (uint32_t)X.hi, vm.D.chain, vm.D.img, S, L, vm.D.sfx, vm.n, [&](int i) { return vm.s[i]; });
  else
    cstd::chain_match(cstd::u128(c0 | ((unsigned long long)c1 << 32), c2), X, vm.D.chain, vm.D.img, S, L, vm.D.sfx, vm.n, [&](int i) { return vm.s[i]; });
  return true;
}
#endif
CS_HD int row_contains_re(cstd::Tdfa& vm, bool anchored) {
  auto none = [](int, int, int) {};
#if !defined(__HIP_DEVICE_COMPILE__)
  if (!anchored) {
    cstd::U128 S, L;
    if (row_chain_host(vm, S, L)) return cstd::u128_any(S) ? 1 : 0;
  }
  if (!anchored && vm.lean_ok()) {  // host builds check the lean scan against the oracle (tests/rowemu)
    uint32_t m0, m1, m2;
    if (vm.has_range2()) vm.build_masks_lean<true>(m0, m1, m2);
    else vm.build_masks_lean<false>(m0, m1, m2);
    const int r = vm.scan_lean_count<cstd::Tdfa::K_CONTAINS>(m0, m1, m2);
    if (r >= 0) return r;
  }
#endif
  return anchored ? vm.scan<cstd::Tdfa::K_MATCH>(0, none) : vm.scan<cstd::Tdfa::K_CONTAINS>(0, none);
}
CS_HD int row_count_re(cstd::Tdfa& vm) {
#if !defined(__HIP_DEVICE_COMPILE__)
  {
    cstd::U128 S, L;
    if (row_chain_host(vm, S, L)) return cstd::u128_popc(S);
  }
  if (vm.lean_ok()) {
    uint32_t m0, m1, m2;
    if (vm.has_range2()) vm.build_masks_lean<true>(m0, m1, m2);
    else vm.build_masks_lean<false>(m0, m1, m2);
    const int r = vm.scan_lean_count<cstd::Tdfa::K_COUNT>(m0, m1, m2);
    if (r >= 0) return r;
  }
#endif
  return vm.scan<cstd::Tdfa::K_COUNT>(0, [](int, int, int) {});
}
// findall on the flat scan loop; emit(k, mb, me) as csvm::row_findall's (its return value is not consulted:
// the callers bound k themselves)
template <class Emit>
CS_HD int row_findall(cstd::Tdfa& vm, Emit&& emit) {
#if !defined(__HIP_DEVICE_COMPILE__)
  {
    cstd::U128 S, L;
    if (row_chain_host(vm, S, L)) {
      int k = 0;
      while (cstd::u128_any(S)) {
        emit(k++, cstd::u128_ctz(S), cstd::u128_ctz(L) + 1);
        S = cstd::u128_clear_lowest(S);
        L = cstd::u128_clear_lowest(L);
      }
      return k;
    }
  }
  if (vm.lean_ok()) {  // host builds check the lean span scan against the oracle (tests/rowemu)
    uint32_t m0, m1, m2;
    if (vm.has_range2()) vm.build_masks_lean<true>(m0, m1, m2);
    else vm.build_masks_lean<false>(m0, m1, m2);
    int sb[128], se[128], k = 0;
    const int r = vm.scan_lean_spans(m0, m1, m2, [&](int mb, int me, int) {
      if (k < 128) {
        sb[k] = mb;
        se[k] = me;
      }
      ++k;
    });
    if (r >= 0 && k <= 128) {
      for (int j = 0; j < k; ++j) emit(j, sb[j], se[j]);
      return k;
    }
  }
#endif
  int k = 0;
  return vm.scan<cstd::Tdfa::K_COUNT>(0, [&](int mb, int me, int) {
    emit(k, mb, me);
    ++k;
  });
}
template <class Emit>
CS_HD void row_replace_matches(cstd::Tdfa& vm, int maxrepl, Emit&& emit) {
  if (maxrepl == 0) return;
#if !defined(__HIP_DEVICE_COMPILE__)
  // host builds (tests/rowemu) route qualifying rows through the lean scan so that it is
  // checked against the oracle with the same fuzz corpus; matches are buffered because
  // a bail-out must leave no trace
  // the unit decomposition of the replace kernels (regex_tdfa.cpp): every unit scanned on its own, in order
  // (header word 31 bit 17: rows with bytes >= 0x80 too -- no NUL --, those bytes taken out of the candidate bits and every
  // unit's scan ending at the unit's end: cs_regex.hip, reclassify_high)
  bool high = false, nul = false;
  for (int i = 0; i < vm.n; ++i) {
    high |= vm.s[i] >= 128;
    nul |= vm.s[i] == 0;
  }
  bool hi_units = high && !nul && ((vm.D.units >> 17) & 3u) && vm.D.nskip > 0 && vm.masks_fit() && vm.D.img[12] <= 4;
  if (hi_units && ((vm.D.units >> 18) & 1u)) {  // (bit 18: every non-ASCII character well formed and outside the builtins' flags)
    const unsigned fmask = (vm.D.units >> 19) & 31u;
    for (int i = 0; i < vm.n && hi_units;) {
      const uint8_t b = vm.s[i];
      if (b < 0x80) {
        ++i;
        continue;
      }
      const unsigned w = csrow::lead_width(b);
      hi_units = w >= 2 && i + (int)w <= vm.n;
      for (unsigned k = 1; k < w && hi_units; ++k) hi_units = csrow::is_cont(vm.s[i + (int)k]);
      if (hi_units) {
        csrow::Char ch;
        csrow::decode_at(vm.s, i, vm.n, ch);
        const unsigned u = csrow::packed_to_cp(ch);
        hi_units = !(u <= 0xFFFFu && (vm.P.flags[u] & fmask) != 0);
      }
      i += (int)w;
    }
  }
  if (maxrepl < 0 && (vm.D.chain >> 16) && cstd::g_chain_host && vm.lean_ok()) {
    // chain patterns on plain-ASCII rows: the matches by marker arithmetic (chain_match; the replace stream kernel's route)
    uint32_t c0, c1, c2;
    if (vm.has_range2()) vm.build_masks_lean<true>(c0, c1, c2);
    else vm.build_masks_lean<false>(c0, c1, c2);
    const unsigned x = (vm.D.units >> 8) & 127u;
    const cstd::U128 R = cstd::u128(c0 | ((unsigned long long)c1 << 32), c2);
    cstd::U128 X = cstd::u128(0, 0), S, L;
    for (int i = 0; x && i < vm.n; ++i)
      if (vm.s[i] == x) {
        if (i < 64) X.lo |= 1ull << i;
        else X.hi |= 1ull << (i - 64);
      }
    if (vm.n <= 95 && cstd::g_chain_host == 1)  // (the three-word form, as the replace stream kernel runs it)
      cstd::chain_match96(c0, c1, c2, (uint32_t)X.lo, (uint32_t)(X.lo >> 32), (uint32_t)X.hi, vm.D.chain, vm.D.img, S, L, vm.D.sfx, vm.n, [&](int i) { return vm.s[i]; });
    else cstd::chain_match(R, X, vm.D.chain, vm.D.img, S, L, vm.D.sfx, vm.n, [&](int i) { return vm.s[i]; });
    while (cstd::u128_any(S)) {
      emit(cstd::u128_ctz(S), cstd::u128_ctz(L) + 1, 1);
      S = cstd::u128_clear_lowest(S);
      L = cstd::u128_clear_lowest(L);
    }
    return;
  }
  if (maxrepl < 0 && (vm.D.units & 1u) && (vm.lean_ok() || hi_units)) {
    int buf[3 * 64];
    int cnt = 0;
    bool bail = false, overflow = false;
    uint32_t c0, c1, c2;
    if (vm.has_range2()) vm.build_masks_lean<true>(c0, c1, c2);
    else vm.build_masks_lean<false>(c0, c1, c2);
    if (hi_units)
      for (int i = 0; i < vm.n; ++i)
        if (vm.s[i] >= 128) {
          if (i < 32) c0 &= ~(1u << i);
          else if (i < 64) c1 &= ~(1u << (i - 32));
          else c2 &= ~(1u << (i - 64));
        }
    const unsigned x = (vm.D.units >> 8) & 127u;
    const cstd::U128 C = cstd::u128(c0 | ((unsigned long long)c1 << 32), c2);
    cstd::U128 X = cstd::u128(0, 0), N;
    for (int i = 0; x && i < vm.n; ++i)
      if (vm.s[i] == x) {
        if (i < 64) X.lo |= 1ull << i;
        else X.hi |= 1ull << (i - 64);
      }
    cstd::U128 W = cstd::unit_ends(C, X, ((vm.D.units >> 16) & 1u) != 0, N);
    while (cstd::u128_any(W) && !bail && !overflow) {
      const int q = cstd::u128_ctz(W);
      W = cstd::u128_clear_lowest(W);
      const cstd::U128 cm = cstd::u128_and(C, cstd::u128_andn(cstd::u128_below(q), cstd::u128_below(cstd::unit_start(N, q))));
      cstd::Tdfa vu(vm.D, vm.P, vm.s, hi_units && q < vm.n ? q : vm.n, vm.sa);
      vu.scan_lean_dispatch(-1, (uint32_t)cm.lo, (uint32_t)(cm.lo >> 32), (uint32_t)cm.hi, [&](int mb, int me, int reps) {
        if (cnt < 64) {
          buf[3 * cnt] = mb;
          buf[3 * cnt + 1] = me;
          buf[3 * cnt + 2] = reps;
          ++cnt;
        } else {
          overflow = true;
        }
      }, bail);
    }
    if (!bail && !overflow) {
      for (int i = 0; i < cnt; ++i) emit(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
      return;
    }
  }
  if (vm.lean_ok()) {
    int buf[3 * 64];
    int cnt = 0;
    bool bail = false, overflow = false;
    uint32_t lm0, lm1, lm2;
    if (vm.has_range2()) vm.build_masks_lean<true>(lm0, lm1, lm2);
    else vm.build_masks_lean<false>(lm0, lm1, lm2);
    vm.scan_lean_dispatch(maxrepl, lm0, lm1, lm2, [&](int mb, int me, int reps) {
      if (cnt < 64) {
        buf[3 * cnt] = mb;
        buf[3 * cnt + 1] = me;
        buf[3 * cnt + 2] = reps;
        ++cnt;
      } else {
        overflow = true;
      }
    }, bail);
    if (!overflow) {
      // the matches found so far are final; a bail-out resumes the generic scan at the round it
      // happened in (the tile kernels do the same)
      for (int i = 0; i < cnt; ++i) emit(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
      if (bail) vm.scan<cstd::Tdfa::K_REPLACE>(maxrepl, emit, vm.lean_resume_from, cnt);
      return;
    }
  }
#endif
  vm.scan<cstd::Tdfa::K_REPLACE>(maxrepl, emit);
}
// ---- the drivers for TdfaWide: the generic scan with eight thread slots (no lean scan, no unit decomposition) ----
CS_HD int row_contains_re(cstd::TdfaWide& vm, bool anchored) {
  auto none = [](int, int, int) {};
  return anchored ? vm.template scan<cstd::Tdfa::K_MATCH, decltype(none)&, cstd::kMaxSlotsWide>(0, none)
                  : vm.template scan<cstd::Tdfa::K_CONTAINS, decltype(none)&, cstd::kMaxSlotsWide>(0, none);
}
CS_HD int row_count_re(cstd::TdfaWide& vm) {
  auto none = [](int, int, int) {};
  return vm.template scan<cstd::Tdfa::K_COUNT, decltype(none)&, cstd::kMaxSlotsWide>(0, none);
}
template <class Emit>
CS_HD int row_findall(cstd::TdfaWide& vm, Emit&& emit) {
  int k = 0;
  auto each = [&](int mb, int me, int) {
    emit(k, mb, me);
    ++k;
  };
  return vm.template scan<cstd::Tdfa::K_COUNT, decltype(each)&, cstd::kMaxSlotsWide>(0, each);
}
template <class Emit>
CS_HD void row_replace_matches(cstd::TdfaWide& vm, int maxrepl, Emit&& emit) {
  if (maxrepl == 0) return;
  vm.template scan<cstd::Tdfa::K_REPLACE, Emit&, cstd::kMaxSlotsWide>(maxrepl, emit);
}
// The match walk of replace_with_backrefs on the DFA: the first matches of the row by the flat scan loop (word-wise
// idle skipping), kept in registers and handed to f(mb, me) only after the scan -- f runs whole DFA passes of its
// own (group_find), which must not sit inside the scan loop -- the rare rest by one find() per match.
template <class F>
CS_HD void walk_matches(cstd::Tdfa& vm, F&& f) {
  constexpr int kKeep = 4;
  int mbs[kKeep], mes[kKeep], cnt = 0;
  row_replace_matches(vm, kKeep, [&](int mb, int me, int) {
#pragma unroll
    for (int j = 0; j < kKeep; ++j)
      if (cnt == j) {
        mbs[j] = mb;
        mes[j] = me;
      }
    ++cnt;
  });
  int last = 0;
#pragma unroll
  for (int j = 0; j < kKeep; ++j)
    if (j < cnt) {
      f(mbs[j], mes[j]);
      last = mes[j];
    }
  if (cnt >= kKeep && mes[kKeep - 1] > mbs[kKeep - 1])
    walk_matches_by_find([&](int from, int& mb, int& me) { return vm.find(from, vm.n, mb, me) > 0; }, f, last);
}
}  // namespace csvm

namespace csrx {
struct Program;
// Host: builds the image; returns an empty vector when the program is not
// convertible within the limits above.  `image` is the list simulator's device
// image of the same program (Program::to_device_image), `flags` the unicode table.
// Shortest number of characters a match of the program consumes (0 = it matches the empty string).
int min_match_chars(const Program& prog);
// `groups_out` (optional) receives the capture-group tag image consumed by cstd::Tdfa::group_find
// (empty when the program has no groups).
std::vector<int32_t> build_tdfa(const Program& prog, const std::vector<int32_t>& image, const uint8_t* flags,
                                std::vector<int32_t>* groups_out = nullptr);
}  // namespace csrx
