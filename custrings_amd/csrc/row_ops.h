// Per-row string logic of the hot path, written in BYTE-OFFSET space over the
// native offsets+chars column (the reference works in character positions on
// its custring_view objects; results are identical for valid UTF-8, which is
// the reference's own input contract).
//
// Everything here is `__host__ __device__` so that the very same functions run
// inside the HIP kernels (kernels.hip) and inside the CPU row-emulation harness
// tests/rowemu (which checks them against oracle/ without a GPU).
//
// Reference semantics restated (paths under /root/reference/cpp/src):
//   UTF-8 helpers      custring_view.inl:48-57,1714-1766 ; util.inl:22-75
//   find / contains    custring_view.inl:481-514 ; strings/find.cu:75-120,237-272
//   replace            strings/modify.cu:109-192
//   strip              custring_view.inl:1398-1598 ; strings/strip.cu
//   lower / upper      strings/case.cu:31-170
//   split              custring_view.inl:1223-1250 ; strings/split.cu:32-87,734-956
//   tokenize           text/tokens.cu:41-121
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CS_HD __host__ __device__ __forceinline__
#else
#define CS_HD inline
#endif

namespace csrow {

typedef uint32_t Char;  // 1-4 raw UTF-8 bytes, big-endian packed

CS_HD bool is_cont(uint8_t b) { return (b & 0xC0) == 0x80; }
// width implied by a lead byte; 0 for a continuation byte
CS_HD unsigned lead_width(uint8_t b) {
  return 1u + ((b & 0xF0) == 0xF0) + ((b & 0xE0) == 0xE0) + ((b & 0xC0) == 0xC0) -
         ((b & 0xC0) == 0x80);
}
// decode the char starting at p[i]; bytes at or past `n` read as 0
CS_HD unsigned decode_at(const uint8_t* p, int i, int n, Char& c) {
  uint8_t b = p[i];
  c = b;
  if (b < 0x80) return 1;
  unsigned w = lead_width(b);
  for (unsigned k = 1; k < w; ++k) c = (c << 8) | ((i + (int)k < n) ? p[i + k] : 0);
  return w;
}
CS_HD unsigned packed_width(Char c) {
  return 1u + ((c & 0xFF00u) > 0) + ((c & 0xFF0000u) > 0) + ((c & 0xFF000000u) > 0);
}
CS_HD unsigned cp_to_packed(unsigned u) {
  if (u < 0x80) return u;
  if (u < 0x800) return ((u << 2) & 0x1F00) | (u & 0x3F) | 0xC080;
  if (u < 0x10000) return ((u << 4) & 0x0F0000) | ((u << 2) & 0x003F00) | (u & 0x3F) | 0xE08080;
  if (u < 0x110000)
    return ((u << 6) & 0x07000000) | ((u << 4) & 0x003F0000) | ((u << 2) & 0x3F00) | (u & 0x3F) |
           0xF0808080u;
  return 0;
}
CS_HD unsigned packed_to_cp(unsigned c) {
  if (c < 0x80) return c;
  if (c < 0xE000) return ((c & 0x1F00) >> 2) | (c & 0x3F);
  if (c < 0xF00000) return ((c & 0x0F0000) >> 4) | ((c & 0x3F00) >> 2) | (c & 0x3F);
  if (c <= 0xF8000000u)  // lead-byte mask 0x03, not 0x07: contract quirk (util.inl:67)
    return ((c & 0x03000000) >> 6) | ((c & 0x3F0000) >> 4) | ((c & 0x3F00) >> 2) | (c & 0x3F);
  return 0;
}
CS_HD int count_chars(const uint8_t* p, int n) {
  int k = 0;
  for (int i = 0; i < n; ++i) k += !is_cont(p[i]);
  return k;
}
// byte offset of character position `chpos` (clamped to n)
CS_HD int byte_of_char(const uint8_t* p, int n, int chpos) {
  int i = 0;
  while (chpos > 0 && i < n) {
    unsigned w = lead_width(p[i]);
    i += w ? (int)w : 1;
    chpos -= w != 0;
  }
  return i < n ? i : n;
}
// advance `k` characters from byte offset i
CS_HD int skip_chars(const uint8_t* p, int n, int i, int k) {
  while (k > 0 && i < n) {
    unsigned w = lead_width(p[i]);
    i += w ? (int)w : 1;
    k -= w != 0;
  }
  return i < n ? i : n;
}

// first byte offset m in [from, to - nb] with p[m..m+nb) == needle; -1 if none
CS_HD int find_bytes(const uint8_t* p, int from, int to, const uint8_t* needle, int nb) {
  for (int m = from; m + nb <= to; ++m) {
    int j = 0;
    while (j < nb && p[m + j] == needle[j]) ++j;
    if (j == nb) return m;
  }
  return -1;
}

// ---- find: char-position window in, char position out -----------------------
CS_HD int row_find(const uint8_t* p, int n, const uint8_t* needle, int nb, int start, int end) {
  if (nb == 0) return -1;  // an empty needle never matches
  if (start < 0) start = 0;
  // the reference turns (start,end) into a count: end - start, negative = rest
  int lo = byte_of_char(p, n, start);
  int hi = n;
  if (end - start >= 0) hi = skip_chars(p, n, lo, end - start);
  int m = find_bytes(p, lo, hi, needle, nb);
  return m < 0 ? -1 : count_chars(p, m);
}

// ---- the rest of the find family (find.cu:36-72, 123-236, 276-387; custring_view.inl:434-442, 481-515, 550-582, 1673-1703) ----
// byte offset of character position `chpos` as custring_view::offset_for_char_pos has it: 0 for 0, the size from the
// character count on (the position is UNSIGNED there: a negative argument is past the end)
CS_HD int offset_for_char_pos(const uint8_t* p, int n, int nchars, unsigned chpos) {
  if (chpos == 0) return 0;
  if (chpos >= (unsigned)nchars) return n;
  return byte_of_char(p, n, (int)chpos);
}
// custring_view::find(str, bytes, pos, count): the window is [pos, pos + count) in characters, count < 0 = the rest
CS_HD int row_find_count(const uint8_t* p, int n, const uint8_t* needle, int nb, unsigned pos, int count) {
  if (nb == 0) return -1;
  const int nchars = count_chars(p, n);
  if (count < 0) count = nchars;
  int end = (int)pos + count;
  if (end < 0 || end > nchars) end = nchars;
  const int spos = offset_for_char_pos(p, n, nchars, pos), epos = offset_for_char_pos(p, n, nchars, (unsigned)end);
  const int m = find_bytes(p, spos, epos, needle, nb);  // (an empty or inverted window: no start fits)
  return m < 0 ? -1 : count_chars(p, m);
}
// custring_view::rfind(str, bytes, pos, count): the LAST occurrence inside the window; the count is taken as it comes
// (a negative one moves the window's end in front of its start unless the sum is negative too)
CS_HD int row_rfind_count(const uint8_t* p, int n, const uint8_t* needle, int nb, unsigned pos, int count) {
  if (nb == 0) return -1;
  const int nchars = count_chars(p, n);
  int end = (int)pos + count;
  if (end < 0 || end > nchars) end = nchars;
  const int spos = offset_for_char_pos(p, n, nchars, pos), epos = offset_for_char_pos(p, n, nchars, (unsigned)end);
  for (int m = epos - nb; m >= spos; --m) {
    int j = 0;
    while (j < nb && p[m + j] == needle[j]) ++j;
    if (j == nb) return count_chars(p, m);
  }
  return -1;
}
// custr::compare (custring.inl:240-261): the difference of the first bytes that differ, else +1 / -1 for the longer side
CS_HD int row_compare(const uint8_t* p, int n, const uint8_t* q, int m) {
  int i = 0;
  for (; i < n && i < m; ++i)
    if (p[i] != q[i]) return (int)p[i] - (int)q[i];
  if (i < n) return 1;
  if (i < m) return -1;
  return 0;
}
CS_HD bool row_starts_with(const uint8_t* p, int n, const uint8_t* q, int m) {
  if (m > n) return false;
  for (int i = 0; i < m; ++i)
    if (p[i] != q[i]) return false;
  return true;
}
CS_HD bool row_ends_with(const uint8_t* p, int n, const uint8_t* q, int m) {
  if (m > n) return false;
  for (int i = 0; i < m; ++i)
    if (p[n - m + i] != q[i]) return false;
  return true;
}

// ---- replace -------------------------------------------------------------------
// maxrepl < 0: unlimited (the reference's cap, nchars, can never bind)
CS_HD int row_replace_size(const uint8_t* p, int n, const uint8_t* needle, int nb, int rb,
                           int maxrepl) {
  int out = n, from = 0;
  for (int left = maxrepl; left != 0; --left) {
    int m = find_bytes(p, from, n, needle, nb);
    if (m < 0) break;
    out += rb - nb;
    from = m + nb;
  }
  return out;
}
CS_HD void row_replace_write(const uint8_t* p, int n, const uint8_t* needle, int nb,
                             const uint8_t* repl, int rb, int maxrepl, uint8_t* o) {
  int from = 0;
  for (int left = maxrepl; left != 0; --left) {
    int m = find_bytes(p, from, n, needle, nb);
    if (m < 0) break;
    for (int i = from; i < m; ++i) *o++ = p[i];
    for (int i = 0; i < rb; ++i) *o++ = repl[i];
    from = m + nb;
  }
  for (int i = from; i < n; ++i) *o++ = p[i];
}

// ---- strip ---------------------------------------------------------------------
struct CharSet {  // a set of characters of any size (custring_view.inl:93-105 walks the caller's string, whatever its length)
  Char c[64];        // the members -- of a set of more than 64 characters: its first 64 non-ASCII members
  int n;             // members in c[]
  uint32_t ascii[4]; // the set's ASCII members as a bitmap (charset_finish): one bit test instead of a walk over c[]
  int wide;          // more than 64 characters: the ASCII members are in the bitmap ONLY, non-ASCII ones beyond c[] in `more`
  const Char* more;  // sorted ascending (device memory in kernels, host memory in the host emulation)
  int nmore;
};
// call once the members are in c[0 .. n)
CS_HD void charset_finish(CharSet& s) {
  s.ascii[0] = s.ascii[1] = s.ascii[2] = s.ascii[3] = 0;
  s.wide = 0;
  s.more = nullptr;
  s.nmore = 0;
  for (int i = 0; i < s.n; ++i)
    if (s.c[i] < 128u) s.ascii[s.c[i] >> 5] |= 1u << (s.c[i] & 31u);
}
CS_HD bool in_set(const CharSet& s, Char ch) {
  if ((s.n > 4 || s.wide) && ch < 128u) {  // (a handful of members: the walk below is cheaper than the word select)
    const unsigned k = ch >> 5;
    const uint32_t w = k == 0 ? s.ascii[0] : (k == 1 ? s.ascii[1] : (k == 2 ? s.ascii[2] : s.ascii[3]));
    return ((w >> (ch & 31u)) & 1u) != 0;
  }
  for (int i = 0; i < s.n; ++i)
    if (s.c[i] == ch) return true;
  int lo = 0, hi = s.nmore;  // (sets beyond 64 non-ASCII characters)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const Char m = s.more[mid];
    if (m == ch) return true;
    if (m < ch) lo = mid + 1;
    else hi = mid;
  }
  return false;
}
// Host: the set of the characters of the UTF-8 string `s` (n bytes).  Up to 64 characters: all of them in c[], as ever.
// Beyond: ASCII members in the bitmap, the first 64 non-ASCII ones in c[], the others sorted into `overflow` -- the caller
// puts them where the kernels can read them and sets `more` / `nmore`.
template <class Vec>
inline CharSet charset_from_utf8(const uint8_t* s, int n, Vec& overflow) {
  CharSet cs;
  cs.n = 0;
  overflow.clear();
  int count = 0;
  for (int i = 0; i < n; ++count) {
    Char c;
    const unsigned w = decode_at(s, i, n, c);
    i += w ? (int)w : 1;
  }
  if (count <= 64) {
    for (int i = 0; i < n;) {
      Char c;
      const unsigned w = decode_at(s, i, n, c);
      cs.c[cs.n++] = c;
      i += w ? (int)w : 1;
    }
    charset_finish(cs);
    return cs;
  }
  uint32_t bits[4] = {0, 0, 0, 0};
  for (int i = 0; i < n;) {
    Char c;
    const unsigned w = decode_at(s, i, n, c);
    i += w ? (int)w : 1;
    if (c < 128u) {
      bits[c >> 5] |= 1u << (c & 31u);
      continue;
    }
    bool seen = false;
    for (int k = 0; k < cs.n && !seen; ++k) seen = cs.c[k] == c;
    if (seen) continue;
    if (cs.n < 64) cs.c[cs.n++] = c;
    else overflow.push_back(c);
  }
  charset_finish(cs);
  for (int k = 0; k < 4; ++k) cs.ascii[k] = bits[k];
  cs.wide = 1;
  for (int a = 1; a < (int)overflow.size(); ++a)  // (insertion sort: a few characters)
    for (int b = a; b > 0 && overflow[b - 1] > overflow[b]; --b) {
      const Char t = overflow[b - 1];
      overflow[b - 1] = overflow[b];
      overflow[b] = t;
    }
  return cs;
}
// side: 0 both, 1 left, 2 right.  Returns [lo,hi) of the kept bytes.
CS_HD void row_strip(const uint8_t* p, int n, const CharSet& set, int side, int& lo, int& hi) {
  lo = 0;
  hi = n;
  if (side != 2) {
    while (lo < n) {
      Char ch;
      unsigned w = decode_at(p, lo, n, ch);
      if (!in_set(set, ch)) break;
      lo += w ? (int)w : 1;
    }
    if (lo >= n) {
      lo = hi = n;
      return;
    }
  }
  if (side != 1) {
    while (hi > lo) {
      int q = hi - 1;
      while (q > lo && is_cont(p[q])) --q;
      Char ch;
      decode_at(p, q, n, ch);
      if (!in_set(set, ch)) break;
      hi = q;
    }
  }
}

// ---- lower / upper -------------------------------------------------------------
// flag bit 32 = upper (lower() maps these), 64 = lower (upper() maps these)
CS_HD int row_case_size(const uint8_t* p, int n, const uint8_t* flags, const uint16_t* cases,
                        unsigned bit) {
  int out = 0;
  for (int i = 0; i < n;) {
    uint8_t b = p[i];
    if (b < 0x80) {  // ASCII maps to ASCII
      ++out;
      ++i;
      continue;
    }
    Char ch;
    unsigned w = decode_at(p, i, n, ch);
    if (w == 0) w = 1;
    unsigned u = packed_to_cp(ch);
    unsigned f = u <= 0xFFFF ? flags[u] : 0;
    out += (f & bit) ? (int)packed_width(cp_to_packed(cases[u])) : (int)packed_width(ch);
    i += (int)w;
  }
  return out;
}
CS_HD void row_case_write(const uint8_t* p, int n, const uint8_t* flags, const uint16_t* cases,
                          unsigned bit, uint8_t* o) {
  for (int i = 0; i < n;) {
    uint8_t b = p[i];
    if (b < 0x80) {
      *o++ = (flags[b] & bit) ? (uint8_t)cases[b] : b;
      ++i;
      continue;
    }
    Char ch;
    unsigned w = decode_at(p, i, n, ch);
    if (w == 0) w = 1;
    unsigned u = packed_to_cp(ch);
    unsigned f = u <= 0xFFFF ? flags[u] : 0;
    if (f & bit) ch = cp_to_packed(cases[u]);
    unsigned ow = packed_width(ch);
    for (unsigned k = 0; k < ow; ++k) *o++ = (uint8_t)(ch >> (8 * (ow - 1 - k)));
    i += (int)w;
  }
}

// ---- split on a delimiter string -------------------------------------------------
// Token COUNT advances the search by `nb` CHARACTERS past the match start (the
// reference adds the delimiter's byte length to a character position,
// custring_view.inl:1243) while token EXTRACTION advances by the delimiter
// itself (split.cu:779-793).  They only differ for multi-byte delimiters.
CS_HD int row_split_count(const uint8_t* p, int n, const uint8_t* d, int nb, int tokens) {
  if (n == 0) return 1;
  int cnt = 1, from = 0;
  for (;;) {
    int m = find_bytes(p, from, n, d, nb);
    if (m < 0) break;
    ++cnt;
    from = skip_chars(p, n, m, nb);
  }
  if (tokens > 0 && cnt > tokens) cnt = tokens;
  return cnt;
}
// Walks the row's `cnt` tokens in order; emit(k, lo, hi) for token k.
template <class Emit>
CS_HD void row_split_tokens(const uint8_t* p, int n, const uint8_t* d, int nb, int cnt,
                            Emit&& emit) {
  int lo = 0;
  for (int k = 0; k < cnt; ++k) {
    int hi = n;
    bool more = k + 1 < cnt;
    int m = more ? find_bytes(p, lo, n, d, nb) : -1;
    if (m >= 0) hi = m;
    emit(k, lo, hi);
    if (more && m < 0) {  // counted more tokens than there are delimiters left
      for (int j = k + 1; j < cnt; ++j) emit(j, lo, n);
      return;
    }
    lo = m + nb;
  }
}

// ---- split / tokenize on whitespace (byte <= ' ') ---------------------------------
CS_HD int row_ws_count(const uint8_t* p, int n) {
  int cnt = 0;
  bool in_tok = false;
  for (int i = 0; i < n; ++i) {
    bool sp = p[i] <= 0x20;
    cnt += (!sp && !in_tok);
    in_tok = !sp;
  }
  return cnt;
}
// split(None,n) column count for a row: natural tokens capped by `tokens`,
// at least 1 (split.cu:52-87)
CS_HD int row_wssplit_count(const uint8_t* p, int n, int tokens) {
  int cnt = row_ws_count(p, n);
  if (tokens > 0 && cnt > tokens) cnt = tokens;
  return cnt == 0 ? 1 : cnt;
}
// emit(k, lo, hi) for each of the row's natural whitespace tokens k < limit;
// when `tokens` > 0 the token with index tokens-1 runs to the end of the row.
template <class Emit>
CS_HD void row_ws_tokens(const uint8_t* p, int n, int tokens, Emit&& emit) {
  int k = 0, i = 0;
  while (i < n) {
    while (i < n && p[i] <= 0x20) ++i;
    if (i >= n) break;
    int lo = i;
    if (tokens > 0 && k == tokens - 1) {
      emit(k, lo, n);
      return;
    }
    while (i < n && p[i] > 0x20) ++i;
    emit(k, lo, i);
    ++k;
  }
}

// ---- rsplit (split.cu:960-1148): the forward token COUNT, tokens located from the right ----
CS_HD int rfind_bytes(const uint8_t* p, int to, const uint8_t* needle, int nb) {  // last occurrence inside [0, to)
  for (int m = to - nb; m >= 0; --m) {
    int j = 0;
    while (j < nb && p[m + j] == needle[j]) ++j;
    if (j == nb) return m;
  }
  return -1;
}
// emit(k, lo, hi) for every column k < cnt (empty tokens included); one walk from the right:
// token k runs from behind the k-th delimiter found from the right to the previous one; when
// the delimiters run out early every remaining column repeats [0, hi) (split.cu:1006-1021)
template <class Emit>
CS_HD void row_rsplit_tokens(const uint8_t* p, int n, const uint8_t* d, int nb, int cnt, Emit&& emit) {
  int hi = n;
  for (int k = cnt - 1; k > 0; --k) {
    const int m = rfind_bytes(p, hi, d, nb);
    if (m < 0) {
      for (int j = k; j > 0; --j) emit(j, 0, hi);
      break;
    }
    emit(k, m + nb, m + nb < hi ? hi : m + nb);
    hi = m;
  }
  emit(0, 0, hi);
}
// Whitespace rsplit, column `col` of a row holding `cnt` columns (split.cu:1098-1124 restated on
// bytes: every byte of a multi-byte character is above ' ').  `ncols` is the column count of the
// whole call, which the reference compares with `tokens`.  false = null in this column.
CS_HD bool row_ws_rtoken(const uint8_t* p, int n, int tokens, int cnt, int ncols, int col, int& lo, int& hi) {
  int c = cnt - 1, spos = 0, epos = n;
  bool spaces = true;
  for (int pos = n; pos > 0; --pos) {
    const bool sp = p[pos - 1] <= 0x20;
    if (spaces == sp) {
      if (spaces) epos = pos - 1;
      else spos = pos - 1;
      continue;
    }
    if (!spaces) {
      spos = 0;
      if (ncols - c == tokens) break;
      spos = pos;
      if (c == col) break;
      epos = pos - 1;
      spos = 0;
      --c;
    }
    spaces = !spaces;
  }
  lo = spos;
  hi = epos;
  return spos < epos;
}

// ---- tokenize on a set of delimiter characters -------------------------------------
template <class Emit>
CS_HD int row_set_tokens(const uint8_t* p, int n, const CharSet& set, Emit&& emit) {
  int k = 0, i = 0;
  while (i < n) {
    Char ch;
    unsigned w = 1;
    // skip delimiters
    while (i < n) {
      w = decode_at(p, i, n, ch);
      if (w == 0) w = 1;
      if (!in_set(set, ch)) break;
      i += (int)w;
    }
    if (i >= n) break;
    int lo = i;
    while (i < n) {
      w = decode_at(p, i, n, ch);
      if (w == 0) w = 1;
      if (in_set(set, ch)) break;
      i += (int)w;
    }
    emit(k, lo, i < n ? i : n);
    ++k;
  }
  return k;
}

}  // namespace csrow
