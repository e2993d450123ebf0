/* NVText -- tokenize + n-grams, hot-path subset of /root/reference/cpp/include/NVText.h
 * over the C ABI.  Header-only. */
#ifndef NVSTRINGS_AMD_NVTEXT_H
#define NVSTRINGS_AMD_NVTEXT_H

#include "NVStrings.h"

class NVText {
 public:
  /* NVText.h:40 -- delimiter nullptr = whitespace, else ANY character of `delimiter` separates */
  static NVStrings* tokenize(NVStrings& strs, const char* delimiter = nullptr) {
    cs_column* c = nullptr;
    NVStrings::check(cs_tokenize(strs.handle(), delimiter, nullptr, &c));
    return NVStrings::adopt(c);
  }
  /* NVText.h:153 */
  static NVStrings* create_ngrams(NVStrings& strs, unsigned int ngrams, const char* separator) {
    cs_column* c = nullptr;
    NVStrings::check(cs_ngrams(strs.handle(), ngrams, separator, nullptr, &c));
    return NVStrings::adopt(c);
  }
};

#endif
