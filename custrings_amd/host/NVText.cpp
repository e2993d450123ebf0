// libNVText.so -- the NVText class (include/nvstrings/NVText.h), out of line, over the C ABI.
#include "nvstrings/NVText.h"

#include "custrings_amd.h"
#include "nvstrings/NVStrings.h"

NVStrings* NVText::tokenize(NVStrings& strs, const char* delimiter) {
  cs_column* c = nullptr;
  NVStrings::check(cs_tokenize(strs.handle(), delimiter, nullptr, &c));
  return NVStrings::adopt(c);
}
NVStrings* NVText::tokenize(NVStrings& strs, NVStrings& delimiters) {
  cs_column* c = nullptr;
  NVStrings::check(cs_tokenize_multi(strs.handle(), delimiters.handle(), nullptr, &c));
  return NVStrings::adopt(c);
}
NVStrings* NVText::unique_tokens(NVStrings& strs, const char* delimiter) {
  cs_column* c = nullptr;
  NVStrings::check(cs_unique_tokens(strs.handle(), delimiter, nullptr, &c));
  return NVStrings::adopt(c);
}
unsigned int NVText::token_count(NVStrings& strs, const char* delimiter, unsigned int* results, bool devmem) {
  NVStrings::check(cs_token_count(strs.handle(), delimiter, results, devmem ? 1 : 0, nullptr));
  return 0;  // tokens.cu:360
}
unsigned int NVText::tokens_counts(NVStrings& strs, NVStrings& tokens, const char* delimiter, unsigned int* results, bool devmem) {
  NVStrings::check(cs_tokens_counts(strs.handle(), tokens.handle(), delimiter, results, devmem ? 1 : 0, nullptr));
  return 0;
}
NVStrings* NVText::replace_tokens(NVStrings& strs, NVStrings& tgts, NVStrings& repls, const char* delimiter) {
  cs_column* c = nullptr;
  NVStrings::check(cs_replace_tokens(strs.handle(), tgts.handle(), repls.handle(), delimiter, nullptr, &c));
  return c ? NVStrings::adopt(c) : nullptr;
}
NVStrings* NVText::normalize_spaces(NVStrings& strs) {
  cs_column* c = nullptr;
  NVStrings::check(cs_normalize_spaces(strs.handle(), nullptr, &c));
  return c ? NVStrings::adopt(c) : nullptr;
}
NVStrings* NVText::create_ngrams(NVStrings& strs, unsigned int ngrams, const char* separator) {
  cs_column* c = nullptr;
  NVStrings::check(cs_ngrams(strs.handle(), ngrams, separator, nullptr, &c));
  return NVStrings::adopt(c);
}
