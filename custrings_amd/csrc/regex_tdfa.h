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
// the threads' START offsets in up to 8 registers: match spans come out
// identical to the list simulation.  Programs that need more than kMaxSlots
// simultaneous threads, more than kMaxStates states or more than 62 distinct
// character predicates are not converted (the caller keeps the list simulator).
//
// Image layout (int32 words):
//   [0] magic 'CSTD' [1] nstates [2] natoms [3] npreds [4] n_nonascii_atoms
//   [5] uses (bit0 word-category, bit1 line-category) [6] off INIT [7] off T1
//   [8] off T2 [9] off preds [10] off atomsig [11] off ACT [12] max slots
//   [13] min match chars [14] off CAT [15] total words
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
//   [19:16] keep (slots j >= keep start at this position; 15 = unchanged)
//   [20] COMPLEX (origins in ACT[entry >> 21], 4 bits per slot, 15 = new)
#pragma once
#include <stdint.h>

#include <vector>

#include "regex_vm.h"

namespace cstd {

constexpr int32_t kMagic = 0x44545343;  // "CSTD"
constexpr int kMaxSlots = 8;
constexpr int kMaxStates = 512;
constexpr int kHeaderWords = 16;
enum { MODE_RESTART = 0, MODE_NORESTART = 1, MODE_SEED_ONCE = 2 };
enum { ATOM_EOT = 0, ATOM_NUL = 1, ATOM_FIRST_CLASS = 2 };
enum { P_CHAR = 0, P_ANY = 1, P_ANYNL = 2, P_CCLASS = 3, P_NCCLASS = 4, P_ISNL = 5, P_ISWORD = 6 };

constexpr uint32_t E_STATE = 0x3FFu, E_STOP = 1u << 10, E_MATCH = 1u << 11, E_COMPLEX = 1u << 20;
CS_HD uint32_t e_match_origin(uint32_t e) { return (e >> 12) & 15u; }
CS_HD uint32_t e_keep(uint32_t e) { return (e >> 16) & 15u; }
constexpr uint32_t E_ACTION = E_MATCH | E_COMPLEX | (15u << 16);  // any bit set / keep != 15 handled below

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
};
CS_HD View make_view(const int32_t* img) {
  View v;
  v.img = img;
  v.nstates = img[1];
  v.natoms = img[2];
  v.npreds = img[3];
  v.nna = img[4];
  v.uses = img[5];
  v.init = (const uint32_t*)(img + img[6]);
  v.t1 = (const uint32_t*)(img + img[7]);
  v.t2 = (const uint32_t*)(img + img[8]);
  v.preds = img + img[9];
  v.atomsig = (const uint32_t*)(img + img[10]);
  v.act = (const uint32_t*)(img + img[11]);
  v.cat = (const uint32_t*)(img + img[14]);
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

  CS_HD Tdfa(const View& d, const csvm::ProgView& p, const uint8_t* row, int bytes) : D(d), P(p), s(row), n(bytes) {}

  CS_HD csrow::Char char_at(int i, unsigned& w) const {
    if (i >= n) {
      w = 1;
      return 0;
    }
    csrow::Char c;
    w = csrow::decode_at(s, i, n, c);
    if (w == 0) w = 1;
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

  // Leftmost-first match whose start lies in [from, win_end); win_end is either
  // the row length (search) or from + 1 (anchored), as in the row drivers.
  CS_HD int find(int from, int win_end, int& mb, int& me) {
    const int mode = (win_end == from + 1) ? MODE_SEED_ONCE : MODE_RESTART;
    uint32_t state = D.init[mode * 8 + prev_cat(from)];
    int st[kMaxSlots];
#pragma unroll
    for (int j = 0; j < kMaxSlots; ++j) st[j] = from;
    int matched = 0;
    int pos = from;
    for (;;) {
      uint32_t e;
      int w = 1;
      if (pos >= n) {
        e = D.t2[state * D.natoms + ATOM_EOT];
      } else {
        uint8_t b = s[pos];
        if (b < 128) {
          e = D.t1[state * 128 + b];
        } else {
          unsigned uw;
          csrow::Char c = char_at(pos, uw);
          w = (int)uw;
          e = D.t2[state * D.natoms + nonascii_atom(c)];
        }
      }
      if (e & E_MATCH) {
        uint32_t o = e_match_origin(e);
        int v = pos;
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j)
          if (o == (uint32_t)j) v = st[j];
        mb = v;
        me = pos;
        matched = 1;
      }
      if (e & E_COMPLEX) {
        uint32_t og = D.act[e >> 21];
        int nst[kMaxSlots];
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j) {
          uint32_t o = (og >> (4 * j)) & 15u;
          int v = pos;
#pragma unroll
          for (int i = 0; i < kMaxSlots; ++i)
            if (o == (uint32_t)i) v = st[i];
          nst[j] = v;
        }
#pragma unroll
        for (int j = 0; j < kMaxSlots; ++j) st[j] = nst[j];
      } else {
        uint32_t keep = e_keep(e);
        if (keep != 15u) {
#pragma unroll
          for (int j = 0; j < kMaxSlots; ++j)
            if ((uint32_t)j >= keep) st[j] = pos;
        }
      }
      if (e & E_STOP) break;
      state = e & E_STATE;
      pos += w;
    }
    return matched;
  }
};

}  // namespace cstd

namespace csrx {
struct Program;
// Host: builds the image; returns an empty vector when the program is not
// convertible within the limits above.  `image` is the list simulator's device
// image of the same program (Program::to_device_image), `flags` the unicode table.
std::vector<int32_t> build_tdfa(const Program& prog, const std::vector<int32_t>& image, const uint8_t* flags);
}  // namespace csrx
