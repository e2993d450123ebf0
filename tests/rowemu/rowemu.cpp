// TEST HARNESS (CPU): runs the product's per-row `__host__ __device__` logic
// (custrings_amd/csrc/row_ops.h, regex_vm.h, regex_compile.cpp) serially on the
// host with the same size-pass / scan / write-pass structure the HIP kernels
// use, so the row logic can be checked against oracle/ in a container without a
// GPU.  It is NOT a CPU fallback of the product: nothing in custrings_amd/
// links it, and the shipped library has no host execution path.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../custrings_amd/csrc/regex_program.h"
#include "../../custrings_amd/csrc/regex_bits.h"
#include "../../custrings_amd/csrc/regex_tdfa.h"
#include "../../custrings_amd/csrc/regex_vm.h"
#include "../../custrings_amd/csrc/row_ops.h"
#include "../../oracle/unicode_tables_gen.h"  // same generated tables the product embeds

using namespace csrow;

struct emu_col {
  int64_t rows = 0;
  std::vector<int64_t> off;
  std::vector<uint8_t> chars;
  std::vector<uint8_t> valid;  // empty = all valid
  bool ok(int64_t r) const { return valid.empty() || ((valid[r >> 3] >> (r & 7)) & 1); }
  const uint8_t* row(int64_t r) const { return chars.data() + off[r]; }
  int len(int64_t r) const { return (int)(off[r + 1] - off[r]); }
};

// sizes (-1 = null) -> column skeleton; then fill(r, dst)
template <class Size, class Fill>
static emu_col* two_pass(int64_t rows, Size size, Fill fill) {
  emu_col* o = new emu_col;
  o->rows = rows;
  o->off.assign(rows + 1, 0);
  std::vector<int> sz(rows);
  bool any_null = false;
  for (int64_t r = 0; r < rows; ++r) {
    sz[r] = size(r);
    any_null |= sz[r] < 0;
    o->off[r + 1] = o->off[r] + (sz[r] < 0 ? 0 : sz[r]);
  }
  o->chars.assign((size_t)o->off[rows] + 16, 0);
  if (any_null) {
    o->valid.assign((rows + 7) / 8, 0);
    for (int64_t r = 0; r < rows; ++r)
      if (sz[r] >= 0) o->valid[r >> 3] |= (uint8_t)(1u << (r & 7));
  }
  for (int64_t r = 0; r < rows; ++r)
    if (sz[r] >= 0) fill(r, o->chars.data() + o->off[r]);
  o->chars.resize((size_t)o->off[rows]);
  return o;
}

static std::vector<Char> g_set_overflow;  // (one set alive at a time: every entry point builds its own)
static CharSet make_set(const char* s) {
  CharSet cs = charset_from_utf8((const uint8_t*)s, (int)strlen(s), g_set_overflow);
  if (!g_set_overflow.empty()) {
    cs.more = g_set_overflow.data();
    cs.nmore = (int)g_set_overflow.size();
  }
  return cs;
}

extern "C" {

emu_col* emu_col_create(int64_t rows, const int64_t* off, const uint8_t* chars, const uint8_t* valid) {
  emu_col* c = new emu_col;
  c->rows = rows;
  c->off.assign(off, off + rows + 1);
  c->chars.assign(chars, chars + off[rows]);
  if (valid) c->valid.assign(valid, valid + (rows + 7) / 8);
  return c;
}
void emu_col_free(emu_col* c) { delete c; }
int64_t emu_col_rows(const emu_col* c) { return c->rows; }
int64_t emu_col_nbytes(const emu_col* c) { return (int64_t)c->chars.size(); }
const int64_t* emu_col_offsets(const emu_col* c) { return c->off.data(); }
const uint8_t* emu_col_chars(const emu_col* c) { return c->chars.data(); }
void emu_col_bitmask(const emu_col* c, uint8_t* out) {
  memset(out, 0, (size_t)((c->rows + 7) / 8));
  for (int64_t r = 0; r < c->rows; ++r)
    if (c->ok(r)) out[r >> 3] |= (uint8_t)(1u << (r & 7));
}
void emu_free(void* p) { free(p); }

static emu_col* change_case(const emu_col* c, unsigned bit) {
  return two_pass(
      c->rows,
      [&](int64_t r) {
        return c->ok(r) ? row_case_size(c->row(r), c->len(r), orc_unicode_flags, orc_charcases, bit) : -1;
      },
      [&](int64_t r, uint8_t* o) {
        row_case_write(c->row(r), c->len(r), orc_unicode_flags, orc_charcases, bit, o);
      });
}
emu_col* emu_lower(const emu_col* c) { return change_case(c, 32); }
emu_col* emu_upper(const emu_col* c) { return change_case(c, 64); }

emu_col* emu_strip(const emu_col* c, const char* to_strip, int side) {
  CharSet set = make_set(to_strip ? to_strip : " \n\t");
  return two_pass(
      c->rows,
      [&](int64_t r) {
        if (!c->ok(r)) return -1;
        int lo, hi;
        row_strip(c->row(r), c->len(r), set, side, lo, hi);
        return hi - lo;
      },
      [&](int64_t r, uint8_t* o) {
        int lo, hi;
        row_strip(c->row(r), c->len(r), set, side, lo, hi);
        memcpy(o, c->row(r) + lo, (size_t)(hi - lo));
      });
}

int64_t emu_find(const emu_col* c, const char* str, int start, int end, int32_t* out) {
  int nb = (int)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) ? row_find(c->row(r), c->len(r), (const uint8_t*)str, nb, start, end) : -2;
    n += out[r] != -1;
  }
  return n;
}
int64_t emu_rfind(const emu_col* c, const char* str, int start, int end, int32_t* out) {
  int nb = (int)strlen(str);
  if (start < 0) start = 0;
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) ? row_rfind_count(c->row(r), c->len(r), (const uint8_t*)str, nb, (unsigned)start, end - start) : -2;
    n += out[r] != -1;
  }
  return n;
}
int64_t emu_find_from(const emu_col* c, const char* str, const int32_t* starts, const int32_t* ends, int32_t* out) {
  int nb = (int)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    int pos = starts ? starts[r] : 0;
    out[r] = c->ok(r) ? row_find_count(c->row(r), c->len(r), (const uint8_t*)str, nb, (unsigned)pos, ends ? ends[r] - pos : -1) : -2;
    n += out[r] != -1;
  }
  return n;
}
int64_t emu_compare(const emu_col* c, const char* str, int32_t* out) {
  int nb = (int)strlen(str);
  if (!nb) return 0;
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) ? row_compare(c->row(r), c->len(r), (const uint8_t*)str, nb) : -1;
    n += out[r] == 0;
  }
  return n;
}
int64_t emu_startswith(const emu_col* c, const char* str, uint8_t* out) {
  int nb = (int)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) && row_starts_with(c->row(r), c->len(r), (const uint8_t*)str, nb);
    n += out[r];
  }
  return n;
}
int64_t emu_endswith(const emu_col* c, const char* str, uint8_t* out) {
  int nb = (int)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) && row_ends_with(c->row(r), c->len(r), (const uint8_t*)str, nb);
    n += out[r];
  }
  return n;
}
int64_t emu_match_strings(const emu_col* a, const emu_col* b, uint8_t* out) {
  if (a->rows != b->rows) return -2;
  int64_t n = 0;
  for (int64_t r = 0; r < a->rows; ++r) {
    bool same = a->ok(r) == b->ok(r);
    if (a->ok(r) && b->ok(r)) same = row_compare(a->row(r), a->len(r), b->row(r), b->len(r)) == 0;
    out[r] = same;
    n += same;
  }
  return n;
}
int64_t emu_find_multiple(const emu_col* c, const emu_col* t, int32_t* out) {
  if (c->rows == 0 || t->rows == 0) return 0;
  for (int64_t r = 0; r < c->rows; ++r)
    for (int64_t j = 0; j < t->rows; ++j)
      out[r * t->rows + j] = c->ok(r) && t->ok(j) ? row_find_count(c->row(r), c->len(r), t->row(j), t->len(j), 0u, -1) : -2;
  int64_t n = 0;
  for (int64_t i = 0; i < c->rows; ++i) n += out[i] != -1;
  return n;
}
int64_t emu_contains(const emu_col* c, const char* str, uint8_t* out) {
  int nb = (int)strlen(str);
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = c->ok(r) && nb > 0 && find_bytes(c->row(r), 0, c->len(r), (const uint8_t*)str, nb) >= 0;
    n += out[r];
  }
  return n;
}
emu_col* emu_replace(const emu_col* c, const char* str, const char* repl, int maxrepl) {
  if (!str || !*str) return nullptr;
  if (!repl) repl = "";
  int nb = (int)strlen(str), rb = (int)strlen(repl);
  return two_pass(
      c->rows,
      [&](int64_t r) {
        return c->ok(r) ? row_replace_size(c->row(r), c->len(r), (const uint8_t*)str, nb, rb, maxrepl) : -1;
      },
      [&](int64_t r, uint8_t* o) {
        row_replace_write(c->row(r), c->len(r), (const uint8_t*)str, nb, (const uint8_t*)repl, rb, maxrepl, o);
      });
}

static int split_any(const emu_col* c, const char* delim, int maxsplit, emu_col*** cols_out, bool reverse);
int emu_split(const emu_col* c, const char* delim, int maxsplit, emu_col*** cols_out) {
  return split_any(c, delim, maxsplit, cols_out, false);
}
// the row-wise rsplit kernels' logic (the product routes the cases that equal split to the split kernels)
int emu_rsplit(const emu_col* c, const char* delim, int maxsplit, emu_col*** cols_out) {
  return split_any(c, delim, maxsplit, cols_out, true);
}
static int split_any(const emu_col* c, const char* delim, int maxsplit, emu_col*** cols_out, bool reverse) {
  int tokens = maxsplit > 0 ? maxsplit + 1 : 0;
  int nb = delim ? (int)strlen(delim) : 0;
  std::vector<int> counts(c->rows, 0);
  int ncols = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->ok(r)) continue;
    counts[r] = delim ? row_split_count(c->row(r), c->len(r), (const uint8_t*)delim, nb, tokens)
                      : row_wssplit_count(c->row(r), c->len(r), tokens);
    ncols = std::max(ncols, counts[r]);
  }
  int nout = ncols ? ncols : 1;
  // all columns in one walk per row: token spans[col][row]
  std::vector<std::vector<int>> lo(nout, std::vector<int>(c->rows, -1)), hi(nout, std::vector<int>(c->rows, -1));
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->ok(r)) continue;
    auto emit = [&](int k, int a, int b) {
      if (k < nout) {
        lo[k][r] = a;
        hi[k][r] = b;
      }
    };
    if (reverse && delim) {
      row_rsplit_tokens(c->row(r), c->len(r), (const uint8_t*)delim, nb, counts[r], emit);
    } else if (reverse) {
      for (int k = 0; k < counts[r]; ++k) {
        int a, b;
        if (row_ws_rtoken(c->row(r), c->len(r), tokens, counts[r], ncols, k, a, b)) emit(k, a, b);
      }
    } else if (delim)
      row_split_tokens(c->row(r), c->len(r), (const uint8_t*)delim, nb, counts[r], emit);
    else
      row_ws_tokens(c->row(r), c->len(r), tokens, emit);
  }
  emu_col** cols = (emu_col**)malloc(sizeof(emu_col*) * nout);
  for (int k = 0; k < nout; ++k)
    cols[k] = two_pass(
        c->rows, [&](int64_t r) { return lo[k][r] < 0 ? -1 : hi[k][r] - lo[k][r]; },
        [&](int64_t r, uint8_t* o) { memcpy(o, c->row(r) + lo[k][r], (size_t)(hi[k][r] - lo[k][r])); });
  *cols_out = cols;
  return nout;
}

// ---- regex ----
struct emu_regex {
  std::vector<int32_t> blob, image, tdfa, gtags, bits;
};
static int g_bits = 0;  // 1: rows the bit-parallel form takes (plain ASCII, at most 95 bytes) go through regex_bits.h
void emu_set_bits(int on) { g_bits = on; }
// [words, classes, flags, alternatives] of the bit-parallel form; 0 words = the program does not convert
void emu_regex_bits_info(const emu_regex* re, int* out) {
  out[0] = (int)re->bits.size();
  out[1] = re->bits.empty() ? 0 : re->bits[1];
  out[2] = re->bits.empty() ? 0 : re->bits[2];
  out[3] = re->bits.empty() ? 0 : re->bits[3];
}
static int g_engine = 1;  // 0 = list simulator only, 1 = tagged DFA when the program converts
void emu_set_engine(int e) { g_engine = e; }
emu_regex* emu_regex_compile(const char* pattern) {
  emu_regex* re = new emu_regex;
  csrx::Program p = csrx::compile(pattern);
  re->blob = p.to_blob();
  re->image = p.to_device_image(orc_unicode_flags);
  re->tdfa = csrx::build_tdfa(p, re->image, orc_unicode_flags, &re->gtags);
  re->bits = csrx::build_bits(p, re->image, orc_unicode_flags);
  return re;
}
// [states, atoms, max slots, min match chars, image words] of the tagged DFA; 0 states = not convertible
void emu_regex_tdfa_info(const emu_regex* re, int* out) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
  out[5] = (int)re->gtags.size();  // capture-group tag image (0: no groups, or not convertible)
  if (re->tdfa.empty()) return;
  out[0] = re->tdfa[1];
  out[1] = re->tdfa[2];
  out[2] = re->tdfa[12];
  out[3] = re->tdfa[13];
  out[4] = (int)re->tdfa.size();
}
// header word 31 of the tagged DFA: the unit decomposition offered to the replace kernels (0: none / not convertible)
int emu_regex_units(const emu_regex* re) { return re->tdfa.empty() ? 0 : re->tdfa[31]; }
// the chain form (regex_tdfa.h: chain_match): items in bits 0..15, their number in bits 16..19 (0: the pattern is no chain)
int emu_regex_chain(const emu_regex* re) { return re->tdfa.empty() ? 0 : (int)cstd::make_view(re->tdfa.data()).chain; }
// the chain's suffix bytes (first byte lowest; their number in bits 20..22 of emu_regex_chain)
// the chain's repetition counts (regex_tdfa.h: chain_item), a byte per item
unsigned long long emu_regex_chain_rep(const emu_regex* re) { return re->tdfa.empty() || !(cstd::make_view(re->tdfa.data()).chain >> 16) ? 0 : cstd::chain_crep(re->tdfa.data()); }
int emu_regex_chain_sfx(const emu_regex* re) { return re->tdfa.empty() ? 0 : (int)cstd::make_view(re->tdfa.data()).sfx; }
// 0: chain patterns keep the unit route in replace_re (both routes are checked against the oracle)
void emu_set_chain(int on) { cstd::g_chain_host = on; }
// adopt a program blob produced elsewhere (e.g. by the real reference compiler)
emu_regex* emu_regex_from_blob(const int32_t* words, int n) {
  emu_regex* re = new emu_regex;
  re->blob.assign(words, words + n);
  return re;
}
void emu_regex_free(emu_regex* re) { delete re; }
int emu_regex_blob(const emu_regex* re, const int32_t** words) {
  *words = re->blob.data();
  return (int)re->blob.size();
}

}  // extern "C"
// the tagged DFA with four thread slots (lean scans, unit decomposition, capture groups) / with five to eight (TdfaWide)
static bool narrow_dfa(const emu_regex* re) { return !re->tdfa.empty() && re->tdfa[12] <= cstd::kMaxSlots; }
template <class F>
static void with_vm(const emu_regex* re, const uint8_t* row, int len, F f) {
  csvm::ProgView P = csvm::make_view(re->image.data(), orc_unicode_flags);
  if (g_engine == 1 && !re->tdfa.empty()) {
    cstd::View D = cstd::make_view(re->tdfa.data());
    if (narrow_dfa(re)) {
      cstd::Tdfa vm(D, P, row, len);
      f(vm);
    } else {
      cstd::TdfaWide vm(D, P, row, len);
      f(vm);
    }
    return;
  }
  std::vector<uint32_t> mem((size_t)csvm::vm_slots(P.ninst) + 1);
  if (P.ninst <= 64) {
    csvm::Vm<true> vm(P, mem.data(), 1, row, len);
    f(vm);
  } else {
    csvm::Vm<false> vm(P, mem.data(), 1, row, len);
    f(vm);
  }
}
// The bit-parallel form on one row, as the stream kernels run it: the row sits somewhere in a staged span (here: behind
// `lead` bytes of something else), every 16-byte piece of the span is classified through the spread table into one bitmap
// per class (regex_bits.h: classify16), the row's class masks are cut out of the bitmaps, the matches follow from the masks.
struct BitsRow {
  cstd::U128 C[csbits::kMaxClasses];
  csbits::View V;
  bool ok = false;
  BitsRow(const emu_regex* re, const uint8_t* row, int len) {
    if (!g_bits || re->bits.empty() || len > csbits::kMaxRowBytes) return;
    for (int i = 0; i < len; ++i)
      if (row[i] == 0 || row[i] >= 0x80) return;
    V = csbits::make_view(re->bits.data());
    uint32_t spread[128];
    for (unsigned c = 0; c < 128; ++c) spread[c] = csbits::spread_entry(((uint32_t)re->bits[csbits::kHeaderWords + (c >> 2)] >> (8 * (c & 3))) & 255u);
    const int lead = (int)(((uintptr_t)row >> 2) % 13) + 3;  // any alignment inside the span
    std::vector<uint8_t> span((size_t)((lead + len + 15 + 16) & ~15), (uint8_t)'z');
    memcpy(span.data() + lead, row, (size_t)len);
    std::vector<uint16_t> bm[csbits::kMaxClasses];
    for (int k = 0; k < V.K; ++k) bm[k].assign(span.size() / 16 + 1, 0);
    for (size_t i = 0; i + 16 <= span.size(); i += 16) {
      uint32_t q[4], pair[4];
      memcpy(q, span.data() + i, 16);
      csbits::classify16(spread, q[0], q[1], q[2], q[3], pair);
      for (int k = 0; k < V.K; ++k) bm[k][i >> 4] = (uint16_t)((pair[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
    }
    for (int k = 0; k < csbits::kMaxClasses; ++k) {
      C[k] = cstd::u128(0, 0);
      for (int i = 0; k < V.K && i < len; ++i) {
        const int p = lead + i;
        if ((bm[k][p >> 4] >> (p & 15)) & 1u) C[k] = cstd::u128_or(C[k], csbits::bit_at(i));
      }
    }
    ok = true;
  }
  auto cls() const {
    return [this](int k, int off) { return csbits::m_from(csbits::shr(C[k], off)); };
  }
  // (the unmasked form: what lies beyond the row is the next row's in the kernels -- here every such bit is set, so that a
  // result that depended on them would show)
  auto raw(int len) const {
    return [this, len](int k, int off) {
      return csbits::m_from(cstd::u128_or(csbits::shr(C[k], off), cstd::u128_andn(cstd::u128(~0ull, ~0ull), cstd::u128_below(len - off > 0 ? len - off : 0))));
    };
  }
};
// the row's matches for replace_re: by the bit-parallel form when it takes the row (no limit on replacements), else by the VM
template <class Emit>
static void replace_matches(const emu_regex* re, const uint8_t* row, int len, int maxrepl, Emit emit) {
  BitsRow b(re, row, len);
  if (b.ok && maxrepl < 0) {
    cstd::U128 S, E;
    csbits::match(b.V, b.cls(), b.raw(len), len, S, E);
    while (cstd::u128_any(S)) {
      const int mb = cstd::u128_ctz(S), me = cstd::u128_ctz(E) + 1;
      S = cstd::u128_clear_lowest(S);
      E = cstd::u128_clear_lowest(E);
      emit(mb, me, 1);
    }
    return;
  }
  with_vm(re, row, len, [&](auto& vm) { csvm::row_replace_matches(vm, maxrepl, emit); });
}
extern "C" {
int64_t emu_contains_re(const emu_col* c, const emu_regex* re, int mode, uint8_t* out) {
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = 0;
    if (c->ok(r)) {
      BitsRow b(re, c->row(r), c->len(r));
      if (b.ok) out[r] = mode ? csbits::match_at_start(b.V, b.cls(), b.raw(c->len(r)), c->len(r)) : csbits::contains(b.V, b.cls(), b.raw(c->len(r)), c->len(r));
      else with_vm(re, c->row(r), c->len(r), [&](auto& vm) { out[r] = (uint8_t)csvm::row_contains_re(vm, mode != 0); });
    }
    n += out[r];
  }
  return n;
}
int64_t emu_count_re(const emu_col* c, const emu_regex* re, int32_t* out) {
  int64_t n = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    out[r] = 0;
    if (c->ok(r)) {
      BitsRow b(re, c->row(r), c->len(r));
      if (b.ok) {
        cstd::U128 S, E;
        csbits::match(b.V, b.cls(), b.raw(c->len(r)), c->len(r), S, E);
        out[r] = cstd::u128_popc(S);
      } else {
        with_vm(re, c->row(r), c->len(r), [&](auto& vm) { out[r] = csvm::row_count_re(vm); });
      }
    }
    n += out[r] > 0;
  }
  return n;
}
emu_col* emu_replace_re(const emu_col* c, const emu_regex* re, const char* repl, int maxrepl) {
  if (!repl) repl = "";
  int rb = (int)strlen(repl);
  return two_pass(
      c->rows,
      [&](int64_t r) {
        if (!c->ok(r)) return -1;
        int out = c->len(r);
        replace_matches(re, c->row(r), c->len(r), maxrepl, [&](int mb, int me, int reps) { out += reps * rb - (me - mb); });
        return out;
      },
      [&](int64_t r, uint8_t* o) {
        const uint8_t* p = c->row(r);
        int copied = 0;
        {
          replace_matches(re, p, c->len(r), maxrepl, [&](int mb, int me, int reps) {
            memcpy(o, p + copied, (size_t)(mb - copied));
            o += mb - copied;
            for (int k = 0; k < reps; ++k) {
              memcpy(o, repl, (size_t)rb);
              o += rb;
            }
            copied = me;
          });
        }
        memcpy(o, p + copied, (size_t)(c->len(r) - copied));
      });
}

// ---- extract (one column per capture group; the kernel's per-row logic) ----
int emu_extract(const emu_col* c, const emu_regex* re, emu_col*** cols_out) {
  csvm::ProgView P = csvm::make_view(re->image.data(), orc_unicode_flags);
  const int groups = re->image[2];
  *cols_out = nullptr;
  if (groups <= 0 || c->rows == 0) return 0;
  std::vector<uint32_t> mem((size_t)csvm::gvm_slots(P.ninst) + 1);
  std::vector<std::vector<int>> lo(groups, std::vector<int>(c->rows, 0)), len(groups, std::vector<int>(c->rows, -1));
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->ok(r)) continue;
    int mb = 0, me = 0;
    bool hit = false;
    with_vm(re, c->row(r), c->len(r), [&](auto& vm) { hit = vm.find(0, vm.n, mb, me) > 0; });
    if (!hit) continue;
    // the kernels' route on rows below 255 bytes: every group of the match from one anchored run, four at a time
    const bool all_at_once = g_engine == 1 && narrow_dfa(re) && !re->gtags.empty() && c->len(r) < 255;
    for (int g0 = 0; all_at_once && g0 < groups; g0 += cstd::Tdfa::kGroupBatch) {
      cstd::View D = cstd::make_view(re->tdfa.data());
      cstd::Tdfa vm(D, P, c->row(r), c->len(r));
      int gb[cstd::Tdfa::kGroupBatch], ge[cstd::Tdfa::kGroupBatch], mend = 0;
      const int cnt = std::min(cstd::Tdfa::kGroupBatch, groups - g0);
      // (the kernels try the backward resolution first; here both forms run wherever the backward one applies and
      // must agree -- the oracle comparison above this library then checks the backward result)
      uint8_t hist_bytes[cstd::Tdfa::kBackSteps];
      int ok = vm.group_find_back(mb, re->gtags.data(), g0 + 1, cnt, gb, ge, mend, cstd::Tdfa::HistBytes{hist_bytes, 1});
      static long long n_back = 0, n_all = 0;  // ROWEMU_STATS=1: how often the backward form applied
      ++(ok < 0 ? n_all : n_back);
      if (getenv("ROWEMU_STATS") && ((n_back + n_all) % 1000) == 0) fprintf(stderr, "rowemu: group runs backward %lld, forward only %lld\n", n_back, n_all);
      {
        int fb[cstd::Tdfa::kGroupBatch], fe[cstd::Tdfa::kGroupBatch], fend = 0;
        const int fok = vm.group_find_all(mb, re->gtags.data(), g0 + 1, cnt, fb, fe, fend);
        if (ok < 0) {
          ok = fok;
          mend = fend;
          for (int k = 0; k < cnt; ++k) gb[k] = fb[k], ge[k] = fe[k];
        } else {
          bool same = ok == fok && (!ok || mend == fend);
          for (int k = 0; ok && k < cnt; ++k) same = same && gb[k] == fb[k] && ge[k] == fe[k];
          if (!same) {
            fprintf(stderr, "rowemu: group_find_back and group_find_all disagree on row %lld\n", (long long)r);
            abort();
          }
        }
      }
      for (int k = 0; k < cnt; ++k)
        if (ok && gb[k] >= 0 && ge[k] > gb[k]) {
          lo[g0 + k][r] = gb[k];
          len[g0 + k][r] = ge[k] - gb[k];
        }
    }
    for (int g = 0; !all_at_once && g < groups; ++g) {
      int x = 0, y = -1;
      bool ok;
      if (g_engine == 1 && narrow_dfa(re) && !re->gtags.empty()) {  // group ranges carried by the tagged DFA
        cstd::View D = cstd::make_view(re->tdfa.data());
        cstd::Tdfa vm(D, P, c->row(r), c->len(r));
        int gb = -1, ge = -1;
        ok = vm.group_find(mb, re->gtags.data(), g + 1, gb, ge) && gb >= 0 && ge > gb;
        x = gb;
        y = ge;
      } else if (P.ninst <= 64) {
        csvm::GroupVm<true> gv(P, mem.data(), 1, c->row(r), c->len(r));
        ok = csvm::row_group_span(gv, mb, g + 1, x, y);
      } else {
        csvm::GroupVm<false> gv(P, mem.data(), 1, c->row(r), c->len(r));
        ok = csvm::row_group_span(gv, mb, g + 1, x, y);
      }
      if (ok) {
        lo[g][r] = x;
        len[g][r] = y - x;
      }
    }
  }
  emu_col** cols = (emu_col**)malloc(sizeof(emu_col*) * groups);
  for (int g = 0; g < groups; ++g)
    cols[g] = two_pass(
        c->rows, [&](int64_t r) { return len[g][r]; },
        [&](int64_t r, uint8_t* o) { memcpy(o, c->row(r) + lo[g][r], (size_t)len[g][r]); });
  *cols_out = cols;
  return groups;
}

// ---- replace_with_backrefs (the kernels' per-row logic; template parsed as the product's host code does) ----
emu_col* emu_replace_with_backrefs(const emu_col* c, const emu_regex* re, const char* repl) {
  csvm::ProgView P = csvm::make_view(re->image.data(), orc_unicode_flags);
  std::string text;
  csvm::BackrefTemplate t{};
  for (const char* p = repl; *p;) {
    if (*p == '\\' && p[1] >= '0' && p[1] <= '9' && t.nrefs < csvm::BackrefTemplate::kMaxRefs) {
      const char* q = p + 1;
      while (*q >= '0' && *q <= '9') ++q;
      t.idx[t.nrefs] = atoi(p + 1);
      t.pos[t.nrefs] = (int)text.size();
      ++t.nrefs;
      p = q;
    } else {
      text.push_back(*p++);
    }
  }
  t.text = (const uint8_t*)text.data();
  t.bytes = (int)text.size();
  t.groups = re->image[2];
  const bool dfa = g_engine == 1 && narrow_dfa(re) && (!re->gtags.empty() || t.groups == 0);
  std::vector<uint32_t> mem((size_t)csvm::gvm_slots(P.ninst) + 1);
  auto run = [&](int64_t r, auto&& out) {
    const uint8_t* p = c->row(r);
    const int n = c->len(r);
    if (dfa) {
      cstd::View D = cstd::make_view(re->tdfa.data());
      cstd::Tdfa vm(D, P, p, n);
      csvm::row_backrefs(
          p, n, t, [&](auto&& f) { csvm::walk_matches(vm, f); },
          [&](int mb, int g, int& x, int& y) {
            return g == 0 ? vm.find(mb, mb + 1, x, y) > 0 : vm.group_find(mb, re->gtags.data(), g, x, y) > 0;
          },
          out);
    } else {
      csvm::row_backrefs(
          p, n, t,
          [&](auto&& f) {
            csvm::walk_matches_by_find(
                [&](int from, int& mb, int& me) {
                  csvm::Vm<false> vm(P, mem.data(), 1, p, n);
                  return vm.find(from, n, mb, me) > 0;
                },
                f);
          },
          [&](int mb, int g, int& x, int& y) {
            if (g == 0) {
              csvm::Vm<false> vm(P, mem.data(), 1, p, n);
              return vm.find(mb, mb + 1, x, y) > 0;
            }
            csvm::GroupVm<false> gv(P, mem.data(), 1, p, n);
            return gv.run(mb, g, x, y) > 0;
          },
          out);
    }
  };
  return two_pass(
      c->rows,
      [&](int64_t r) {
        if (!c->ok(r)) return -1;
        int len = 0;
        run(r, [&](const uint8_t*, int k) { len += k; });
        return len;
      },
      [&](int64_t r, uint8_t* o) {
        run(r, [&](const uint8_t* q, int k) {
          memcpy(o, q, (size_t)k);
          o += k;
        });
      });
}

// ---- findall (column k = every row's k-th match) ----
int emu_findall(const emu_col* c, const emu_regex* re, emu_col*** cols_out) {
  *cols_out = nullptr;
  if (c->rows == 0) return 0;
  std::vector<std::vector<std::pair<int, int>>> spans(c->rows);
  int ncols = 0;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->ok(r)) continue;
    with_vm(re, c->row(r), c->len(r), [&](auto& vm) {
      csvm::row_findall(vm, [&](int, int mb, int me) {
        spans[r].push_back({mb, me});
        return true;
      });
    });
    ncols = std::max(ncols, (int)spans[r].size());
  }
  int nout = ncols ? ncols : 1;
  emu_col** cols = (emu_col**)malloc(sizeof(emu_col*) * nout);
  for (int k = 0; k < nout; ++k)
    cols[k] = two_pass(
        c->rows, [&](int64_t r) { return k < (int)spans[r].size() ? spans[r][k].second - spans[r][k].first : -1; },
        [&](int64_t r, uint8_t* o) { memcpy(o, c->row(r) + spans[r][k].first, (size_t)(spans[r][k].second - spans[r][k].first)); });
  *cols_out = cols;
  return nout;
}

// ---- tokenize ----
emu_col* emu_tokenize(const emu_col* c, const char* delim) {
  CharSet set = make_set(delim ? delim : "");
  std::vector<int64_t> src;
  std::vector<int> lo, hi;
  for (int64_t r = 0; r < c->rows; ++r) {
    if (!c->ok(r)) continue;
    auto emit = [&](int, int a, int b) {
      src.push_back(r);
      lo.push_back(a);
      hi.push_back(b);
    };
    if (delim)
      row_set_tokens(c->row(r), c->len(r), set, emit);
    else
      row_ws_tokens(c->row(r), c->len(r), 0, emit);
  }
  return two_pass(
      (int64_t)src.size(), [&](int64_t t) { return hi[t] - lo[t]; },
      [&](int64_t t, uint8_t* o) { memcpy(o, c->row(src[t]) + lo[t], (size_t)(hi[t] - lo[t])); });
}

}  // extern "C"
