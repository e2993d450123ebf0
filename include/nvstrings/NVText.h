/* NVText -- tokenize, n-grams and the token counters of /root/reference/cpp/include/NVText.h over the C ABI.
 * Out-of-line, exported by libNVText.so (custrings_amd/host/NVText.cpp) under the reference's mangled names. */
#ifndef NVSTRINGS_AMD_NVTEXT_H
#define NVSTRINGS_AMD_NVTEXT_H

class NVStrings;

class NVText {
 public:
  /* NVText.h:40 -- delimiter nullptr = whitespace, else ANY character of `delimiter` separates */
  static NVStrings* tokenize(NVStrings& strs, const char* delimiter = nullptr);
  /* NVText.h:48 -- every row of `delimiters` is a whole-string delimiter */
  static NVStrings* tokenize(NVStrings& strs, NVStrings& delimiters);
  /* NVText.h:56-116 (tokens.cu:262-716) */
  static NVStrings* unique_tokens(NVStrings& strs, const char* delimiter = nullptr);
  static unsigned int token_count(NVStrings& strs, const char* delimiter, unsigned int* results, bool devmem = true);
  static unsigned int tokens_counts(NVStrings& strs, NVStrings& tokens, const char* delimiter, unsigned int* results, bool devmem = true);
  static NVStrings* replace_tokens(NVStrings& strs, NVStrings& tgts, NVStrings& repls, const char* delimiter = nullptr);
  static NVStrings* normalize_spaces(NVStrings& strs);
  /* NVText.h:153 */
  static NVStrings* create_ngrams(NVStrings& strs, unsigned int ngrams, const char* separator);
};

#endif
