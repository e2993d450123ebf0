/* cs_synth_spec.h -- definition of the synthetic benchmark columns
 * (BASELINE.md section 3).  Row r of a column is a pure function of
 * (kind, seed, r, param); integer arithmetic only, so the device generator
 * (custrings_amd/csrc/cs_synth.hip) and the CPU generator (oracle/) produce
 * identical bytes.  This header is a workload SPECIFICATION shared by both; it
 * contains no string-op logic.
 *
 *  kind 2 (C2): 64-char rows of ASCII words (len 2-9, mixed case) separated by
 *          single spaces; 25 % rows have a leading and a trailing space; 5 %
 *          rows carry 1-3 two-byte chars (e-acute, lower/upper) so they are
 *          65-67 bytes; 1 % null; 0.5 % empty.
 *  kind 3 (C3 / headline): log lines "<METHOD> /<path> [<ip> ][<ip> ]<status>
 *          <words...>", 48-80 bytes (uniform target, stretched when the fixed
 *          prefix is longer); 50 % rows hold one dotted quad, 5 % two, 45 %
 *          none; no nulls.
 *  kind 4 (C4): exactly 16 ASCII chars naming one of `param` (K) distinct
 *          tokens drawn log-uniformly (Zipf-like, s ~ 1); 0.1 % null.
 *  kind 5 (C5): tweet-like text, 40-150 bytes of words (len 1-9) with single
 *          spaces, 0.6 % rows with one two-byte char, 0.5 % null.
 */
#ifndef CS_SYNTH_SPEC_H
#define CS_SYNTH_SPEC_H
#include <stdint.h>

#if defined(__HIPCC__)
#define CS_SYNTH_HD __host__ __device__ inline
#else
#define CS_SYNTH_HD static inline
#endif

#define CS_SYNTH_SEED 20240607ull

typedef struct cs_rng { uint64_t s; } cs_rng;
CS_SYNTH_HD uint64_t cs_rng_next(cs_rng* r) { /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
CS_SYNTH_HD uint64_t cs_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* The row index is hashed (not just scaled) so that neighbouring rows do not
 * get shifted copies of one splitmix64 stream. */
CS_SYNTH_HD cs_rng cs_rng_for_row(uint64_t seed, int64_t row, int kind) {
  cs_rng r;
  r.s = cs_mix64(seed * 0xD1342543DE82EF95ull + (uint64_t)kind) ^
        cs_mix64((uint64_t)row * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull);
  cs_rng_next(&r);
  return r;
}
CS_SYNTH_HD uint32_t cs_rng_below(cs_rng* r, uint32_t n) { /* n < 2^31 */
  return (uint32_t)(((cs_rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

typedef struct cs_sink { uint8_t* dst; int n; } cs_sink; /* dst NULL = count only */
CS_SYNTH_HD void cs_put(cs_sink* s, uint8_t b) {
  if (s->dst) s->dst[s->n] = b;
  s->n++;
}
CS_SYNTH_HD void cs_put_uint(cs_sink* s, uint32_t v) {
  if (v >= 100) cs_put(s, (uint8_t)('0' + v / 100));
  if (v >= 10) cs_put(s, (uint8_t)('0' + (v / 10) % 10));
  cs_put(s, (uint8_t)('0' + v % 10));
}
/* words of 2..9 (or lo..hi) letters and single spaces until exactly `target` bytes */
CS_SYNTH_HD void cs_fill_words(cs_sink* s, cs_rng* r, int target, int lo, int hi, int mixed_case) {
  while (s->n < target) {
    int w = lo + (int)cs_rng_below(r, (uint32_t)(hi - lo + 1));
    uint64_t bits = cs_rng_next(r);
    for (int k = 0; k < w && s->n < target; ++k) {
      uint8_t ch = (uint8_t)('a' + (bits % 26));
      bits /= 26;
      if (mixed_case && (bits & 1)) ch = (uint8_t)(ch - 32);
      bits >>= 1;
      cs_put(s, ch);
    }
    if (s->n < target - 1) cs_put(s, ' ');
    else if (s->n < target) cs_put(s, 'x');
  }
}

/* Returns 1 when the row is null. */
CS_SYNTH_HD int cs_synth_is_null(int kind, uint64_t seed, int64_t row) {
  cs_rng r = cs_rng_for_row(seed ^ 0xA5A5A5A5ull, row, kind);
  uint32_t u = cs_rng_below(&r, 100000u);
  if (kind == 2) return u < 1000u;
  if (kind == 4) return u < 100u;
  if (kind == 5) return u < 500u;
  return 0;
}

/* Writes row bytes to dst (may be NULL) and returns the byte length. */
CS_SYNTH_HD int cs_synth_row(int kind, uint64_t seed, int64_t row, int64_t param, uint8_t* dst) {
  cs_rng r = cs_rng_for_row(seed, row, kind);
  cs_sink s;
  s.dst = dst;
  s.n = 0;
  if (kind == 2) {
    uint32_t u = cs_rng_below(&r, 1000u);
    if (u < 5u) return 0; /* empty */
    int pad = cs_rng_below(&r, 100u) < 25u;
    int accents = cs_rng_below(&r, 100u) < 5u ? 1 + (int)cs_rng_below(&r, 3u) : 0;
    /* 64 characters: accents are inserted as whole 2-byte chars at word starts */
    int chars = 64, target_bytes = 64 + accents;
    if (pad) cs_put(&s, ' ');
    int body_end = target_bytes - (pad ? 1 : 0);
    (void)chars;
    while (s.n < body_end) {
      if (accents > 0 && s.n + 2 <= body_end) {
        cs_put(&s, 0xC3);
        cs_put(&s, (cs_rng_next(&r) & 1) ? 0xA9 : 0x89);
        --accents;
      }
      int w = 2 + (int)cs_rng_below(&r, 8u);
      uint64_t bits = cs_rng_next(&r);
      for (int k = 0; k < w && s.n < body_end; ++k) {
        uint8_t ch = (uint8_t)('a' + (bits % 26));
        bits /= 26;
        if (bits & 1) ch = (uint8_t)(ch - 32);
        bits >>= 1;
        cs_put(&s, ch);
      }
      if (s.n < body_end - 1) cs_put(&s, ' ');
      else if (s.n < body_end) cs_put(&s, 'x');
    }
    if (pad) cs_put(&s, ' ');
    return s.n;
  }
  if (kind == 3) {
    int target = 48 + (int)cs_rng_below(&r, 33u);
    uint32_t m = cs_rng_below(&r, 5u);
    const char* meth = m == 0 ? "GET" : m == 1 ? "POST" : m == 2 ? "PUT" : m == 3 ? "DELETE" : "HEAD";
    for (const char* q = meth; *q; ++q) cs_put(&s, (uint8_t)*q);
    cs_put(&s, ' ');
    cs_put(&s, '/');
    {
      int plen = 5 + (int)cs_rng_below(&r, 9u);
      uint64_t bits = cs_rng_next(&r);
      for (int k = 0; k < plen; ++k) {
        uint32_t v = (uint32_t)(bits % 27);
        bits /= 27;
        cs_put(&s, v == 26 && k > 0 && k < plen - 1 ? '/' : (uint8_t)('a' + v % 26));
      }
    }
    cs_put(&s, ' ');
    uint32_t ipsel = cs_rng_below(&r, 100u);
    int nip = ipsel < 50u ? 1 : (ipsel < 55u ? 2 : 0);
    for (int k = 0; k < nip; ++k) {
      uint64_t bits = cs_rng_next(&r);
      for (int o = 0; o < 4; ++o) {
        cs_put_uint(&s, (uint32_t)((bits >> (8 * o)) & 255u));
        if (o < 3) cs_put(&s, '.');
      }
      cs_put(&s, ' ');
    }
    {
      static const uint16_t codes[8] = {200, 200, 200, 301, 304, 404, 500, 503};
      cs_put_uint(&s, codes[cs_rng_below(&r, 8u)]);
    }
    if (s.n < target) {
      cs_put(&s, ' ');
      cs_fill_words(&s, &r, target, 2, 9, 0);
    }
    return s.n;
  }
  if (kind == 4) {
    uint64_t K = param > 0 ? (uint64_t)param : 1u;
    /* log-uniform token id in [0, K): level L uniform, id uniform in [2^L, 2^(L+1)) */
    int levels = 0;
    while ((1ull << levels) < K && levels < 62) ++levels;
    uint64_t id;
    if (levels == 0) {
      id = 0;
    } else {
      int L = (int)cs_rng_below(&r, (uint32_t)levels);
      uint64_t lo = 1ull << L, span = lo;
      id = lo + (cs_rng_next(&r) % span) - 1;
      if (id >= K) id = id % K;
    }
    cs_rng t;
    t.s = id * 0x9E3779B97F4A7C15ull + 12345u;
    uint64_t a = cs_rng_next(&t), b = cs_rng_next(&t);
    for (int k = 0; k < 16; ++k) {
      uint64_t* src = k < 8 ? &a : &b;
      uint32_t v = (uint32_t)(*src % 36);
      *src /= 36;
      cs_put(&s, (uint8_t)(v < 10 ? '0' + v : 'a' + (v - 10)));
    }
    return s.n;
  }
  if (kind == 5) {
    int target = 40 + (int)cs_rng_below(&r, 111u);
    if (cs_rng_below(&r, 1000u) < 6u) {
      cs_put(&s, 0xC3);
      cs_put(&s, 0xA9);
    }
    cs_fill_words(&s, &r, target, 1, 9, 1);
    return s.n;
  }
  return 0;
}

/* ---- column digest (full-size parity: "checksum of checksums") ---------------
 * digest(column) = sum over rows (mod 2^64) of cs_digest_row(r, bytes, n, valid):
 * FNV-1a of the row bytes, mixed with the row index and length, so it is
 * sensitive to content, order, offsets and validity, and can be accumulated in
 * any order (device atomics). */
CS_SYNTH_HD uint64_t cs_digest_row(uint64_t row, const uint8_t* p, int n, int valid) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (int i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  h ^= (uint64_t)n * 0x9E3779B97F4A7C15ull;
  if (!valid) h = 0x6c6c756e5f5f5f5full;
  return cs_mix64(h + cs_mix64(row + 1));
}

#endif /* CS_SYNTH_SPEC_H */
