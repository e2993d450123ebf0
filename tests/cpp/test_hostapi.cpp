// Mirrors the reference's gtests for the hot path against the C++ host classes
// (include/nvstrings/*.h), LINKED AGAINST libNVStrings.so / libNVCategory.so / libNVText.so ONLY -- the
// libraries a consumer of the reference links -- never against the C ABI directly: cpp/tests/test_split.cpp:10-46, test_extract.cpp:10-25, test_replace.cpp:16-52,
// test_count.cu:11-101, test_strip.cpp:8-32, test_case.cpp:9-27, test_find.cu:25-76,
// test_text.cu:15-26, python/tests/test_category.py:33-45.
//   test_hostapi nogpu  -> checks the no-device error path only
//   test_hostapi        -> runs the known-answer tests on the GPU
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "nvstrings/NVCategory.h"
#include "nvstrings/NVStrings.h"
#include "nvstrings/NVText.h"

static int failures = 0;
#define EXPECT(c)                                                   \
  do {                                                              \
    if (!(c)) {                                                     \
      ++failures;                                                   \
      printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c);         \
    }                                                               \
  } while (0)

// cpp/tests/utils.h:7-45 (verify_strings): byte_count + to_host, null <=> nullptr
static bool verify_strings(NVStrings* d, const std::vector<const char*>& expected) {
  unsigned count = d->size();
  if (count != expected.size()) return false;
  std::vector<int> lengths(count);
  d->byte_count(lengths.data(), false);
  std::vector<std::string> bufs(count);
  std::vector<char*> ptrs(count);
  for (unsigned i = 0; i < count; ++i) {
    bufs[i].assign(lengths[i] > 0 ? (size_t)lengths[i] : 0, '\0');
    ptrs[i] = lengths[i] < 0 ? nullptr : (bufs[i].empty() ? (char*)"" : &bufs[i][0]);
    if (lengths[i] == 0) ptrs[i] = nullptr;  // nothing to copy
  }
  d->to_host(ptrs.data(), 0, (int)count);
  for (unsigned i = 0; i < count; ++i) {
    if ((lengths[i] < 0) != (expected[i] == nullptr)) return false;
    if (expected[i] && bufs[i] != expected[i]) return false;
  }
  return true;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "nogpu")) {
    const char* one[] = {"a"};
    try {
      NVStrings* s = NVStrings::create_from_array(one, 1);
      printf("a device is present (%u row)\n", s->size());
      NVStrings::destroy(s);
    } catch (const std::runtime_error& e) {
      printf("ok: %s\n", e.what());
    }
    return 0;
  }
  {  // split
    std::vector<const char*> h{"Héllo thesé", nullptr, "are some", "tést String", ""};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    std::vector<NVStrings*> r;
    EXPECT(strs->split(-1, r) == 2);
    EXPECT(verify_strings(r[0], {"Héllo", nullptr, "are", "tést", nullptr}));
    EXPECT(verify_strings(r[1], {"thesé", nullptr, "some", "String", nullptr}));
    for (auto* p : r) NVStrings::destroy(p);
    r.clear();
    EXPECT(strs->split("s", -1, r) == 2);
    EXPECT(verify_strings(r[0], {"Héllo the", nullptr, "are ", "té", ""}));
    EXPECT(verify_strings(r[1], {"é", nullptr, "ome", "t String", nullptr}));
    for (auto* p : r) NVStrings::destroy(p);
    NVStrings::destroy(strs);
  }
  {  // extract (cpp/tests/test_extract.cpp:10-25)
    std::vector<const char*> h{"First Last", "Joe Schmoe", "John Smith", "Jane Smith", "Beyonce", "Sting", nullptr, ""};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    std::vector<NVStrings*> r;
    EXPECT(strs->extract("(\\w+) (\\w+)", r) == 2);
    EXPECT(r.size() == 2);
    if (r.size() == 2) {
      EXPECT(verify_strings(r[0], {"First", "Joe", "John", "Jane", nullptr, nullptr, nullptr, nullptr}));
      EXPECT(verify_strings(r[1], {"Last", "Schmoe", "Smith", "Smith", nullptr, nullptr, nullptr, nullptr}));
    }
    for (auto* p : r) NVStrings::destroy(p);
    r.clear();
    EXPECT(strs->extract_record("(\\w+) (\\w+)", r) == 8);  // cpp/tests/test_extract.cpp:27-52
    if (r.size() == 8) {
      EXPECT(verify_strings(r[0], {"First", "Last"}));
      EXPECT(verify_strings(r[3], {"Jane", "Smith"}));
      EXPECT(verify_strings(r[4], {nullptr, nullptr}));
      EXPECT(verify_strings(r[6], {nullptr, nullptr}));
    }
    for (auto* p : r) NVStrings::destroy(p);
    NVStrings::destroy(strs);
  }
  {  // replace / replace_re
    std::vector<const char*> h{"the quick brown fox jumps over the lazy dog",
                               "the fat cat lays next to the other accénted cat",
                               "a slow moving turtlé cannot catch the bird",
                               "which can be composéd together to form a more complete",
                               "thé result does not include the value in the sum in",
                               "", "absent stop words"};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    NVStrings* got = strs->replace("the ", "++++ ");
    EXPECT(verify_strings(got, {"++++ quick brown fox jumps over ++++ lazy dog",
                                "++++ fat cat lays next to ++++ other accénted cat",
                                "a slow moving turtlé cannot catch ++++ bird",
                                "which can be composéd together to form a more complete",
                                "thé result does not include ++++ value in ++++ sum in", "", "absent stop words"}));
    NVStrings::destroy(got);
    got = strs->replace_re("(\\bin\\b)|(\\ba\\b)|(\\bthe\\b)", "=");
    EXPECT(verify_strings(got, {"= quick brown fox jumps over = lazy dog",
                                "= fat cat lays next to = other accénted cat",
                                "= slow moving turtlé cannot catch = bird",
                                "which can be composéd together to form = more complete",
                                "thé result does not include = value = = sum =", "", "absent stop words"}));
    NVStrings::destroy(got);
    bool threw = false;
    try {
      strs->replace_re("", "x");
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    EXPECT(threw);
    NVStrings::destroy(strs);
  }
  {  // contains / contains_re / match / count_re
    std::vector<const char*> h{"The quick brown @fox jumps", "ovér the", "lazy @dog", "1234", "00:0:00", nullptr, ""};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    bool res[7];
    auto same = [&](std::initializer_list<bool> e) {
      int i = 0;
      for (bool b : e)
        if (res[i++] != b) return false;
      return true;
    };
    strs->contains("é", res, false);
    EXPECT(same({false, true, false, false, false, false, false}));
    strs->contains_re("\\d+", res, false);
    EXPECT(same({false, false, false, true, true, false, false}));
    strs->contains_re("@\\w+", res, false);
    EXPECT(same({true, false, true, false, false, false, false}));
    strs->match("ov[eé]r", res, false);
    EXPECT(same({false, true, false, false, false, false, false}));
    strs->match("[tT]he", res, false);
    EXPECT(same({true, false, false, false, false, false, false}));
    int cnt[7];
    strs->count_re("\\d+:\\d+", cnt, false);
    EXPECT(cnt[4] == 1 && cnt[0] == 0 && cnt[3] == 0 && cnt[5] == 0);
    EXPECT(strs->contains_re(nullptr, res, false) == -1);
    NVStrings::destroy(strs);
  }
  {  // strip, case, find
    std::vector<const char*> h{" hello  ", "   thesé ", nullptr, "ARE THE", " tést  strings ", ""};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    NVStrings* got = strs->lstrip(" ");
    EXPECT(verify_strings(got, {"hello  ", "thesé ", nullptr, "ARE THE", "tést  strings ", ""}));
    NVStrings::destroy(got);
    got = strs->rstrip(" ");
    EXPECT(verify_strings(got, {" hello", "   thesé", nullptr, "ARE THE", " tést  strings", ""}));
    NVStrings::destroy(got);
    got = strs->strip(" ");
    EXPECT(verify_strings(got, {"hello", "thesé", nullptr, "ARE THE", "tést  strings", ""}));
    NVStrings::destroy(got);
    NVStrings::destroy(strs);
    std::vector<const char*> c{"Examples aBc", "thesé", nullptr, "ARE THE", "tést strings", ""};
    strs = NVStrings::create_from_array(c.data(), c.size());
    got = strs->lower();
    EXPECT(verify_strings(got, {"examples abc", "thesé", nullptr, "are the", "tést strings", ""}));
    NVStrings::destroy(got);
    got = strs->upper();
    EXPECT(verify_strings(got, {"EXAMPLES ABC", "THESÉ", nullptr, "ARE THE", "TÉST STRINGS", ""}));
    NVStrings::destroy(got);
    NVStrings::destroy(strs);
    std::vector<const char*> f{"Héllo", "thesé", nullptr, "ARE THE", "tést strings", ""};
    strs = NVStrings::create_from_array(f.data(), f.size());
    int pos[6];
    strs->find("é", 0, -1, pos, false);
    int e[] = {1, 4, -2, -1, 1, -1};
    for (int i = 0; i < 6; ++i) EXPECT(pos[i] == e[i]);
    NVStrings::destroy(strs);
  }
  {  // tokenize, ngrams, category
    std::vector<const char*> t{"the fox jumped over the dog", "the dog chased the cat", "the cat chased the mouse",
                               nullptr, "", "the mouse ate the cheese"};
    NVStrings* strs = NVStrings::create_from_array(t.data(), t.size());
    NVStrings* tok = NVText::tokenize(*strs);
    EXPECT(verify_strings(tok, {"the", "fox", "jumped", "over", "the", "dog", "the", "dog", "chased", "the", "cat",
                                "the", "cat", "chased", "the", "mouse", "the", "mouse", "ate", "the", "cheese"}));
    NVStrings* bi = NVText::create_ngrams(*tok, 2, "_");
    EXPECT(bi->size() == 20);
    NVStrings::destroy(bi);
    NVStrings::destroy(tok);
    NVStrings::destroy(strs);
    std::vector<const char*> e{"eee", "aaa", "eee", "ddd", "ccc", "ccc", "ccc", "eee", "aaa"};
    strs = NVStrings::create_from_array(e.data(), e.size());
    NVCategory* cat = NVCategory::create_from_strings(*strs);
    EXPECT(cat->size() == 9 && cat->keys_size() == 4);
    NVStrings* keys = cat->get_keys();
    EXPECT(verify_strings(keys, {"aaa", "ccc", "ddd", "eee"}));
    int vals[9], ev[] = {3, 0, 3, 2, 1, 1, 1, 3, 0};
    cat->get_values(vals, false);
    for (int i = 0; i < 9; ++i) EXPECT(vals[i] == ev[i]);
    NVStrings::destroy(keys);
    NVCategory::destroy(cat);
    NVStrings::destroy(strs);
  }
  {  // array / combine / records / partition (cpp/tests/test_array.cu, test_combine.cpp, test_split.cpp:61-198)
    std::vector<const char*> h{"John Smith", "Joe Blow", "Jane Smith", nullptr, ""};
    NVStrings* strs = NVStrings::create_from_array(h.data(), h.size());
    NVStrings* got = strs->sublist(1, 4);
    EXPECT(verify_strings(got, {"Joe Blow", "Jane Smith", nullptr}));
    NVStrings::destroy(got);
    int idx[] = {1, 3, 2};
    got = strs->gather(idx, 3, false);
    EXPECT(verify_strings(got, {"Joe Blow", nullptr, "Jane Smith"}));
    NVStrings::destroy(got);
    bool threw = false;
    int bad[] = {0, 5};
    try {
      strs->gather(bad, 2, false);
    } catch (const std::out_of_range&) {
      threw = true;
    }
    EXPECT(threw);
    bool mask[] = {true, false, false, false, true};
    got = strs->gather(mask, false);
    EXPECT(verify_strings(got, {"John Smith", ""}));
    NVStrings::destroy(got);
    got = strs->sort(NVStrings::length);
    EXPECT(verify_strings(got, {nullptr, "", "Joe Blow", "John Smith", "Jane Smith"}));
    NVStrings::destroy(got);
    unsigned int ord[5], eo[] = {0, 1, 2, 4, 3};
    strs->order(NVStrings::name, false, ord, false, false);
    for (int i = 0; i < 5; ++i) EXPECT(ord[i] == eo[i]);
    int lens[5], el[] = {10, 8, 10, -1, 0};
    strs->len(lens, false);
    for (int i = 0; i < 5; ++i) EXPECT(lens[i] == el[i]);
    NVStrings* j = strs->join(":", "_");
    EXPECT(verify_strings(j, {"John Smith:Joe Blow:Jane Smith:_:"}));
    NVStrings::destroy(j);
    NVStrings* c2 = strs->cat(strs, "-", "?");
    EXPECT(verify_strings(c2, {"John Smith-John Smith", "Joe Blow-Joe Blow", "Jane Smith-Jane Smith", "?-?", "-"}));
    NVStrings::destroy(c2);
    // create_index -> create_from_index round trip (device pointers into the instance)
    std::vector<std::pair<const char*, size_t>> pairs(5);
    strs->create_index(pairs.data(), false);
    EXPECT(pairs[3].first == nullptr && pairs[0].second == 10);
    NVStrings* again = NVStrings::create_from_index(pairs.data(), 5, false);
    EXPECT(verify_strings(again, {"John Smith", "Joe Blow", "Jane Smith", nullptr, ""}));
    NVStrings::destroy(again);
    NVStrings::destroy(strs);
    std::vector<const char*> sp{"Héllo thesé", nullptr, "are some", "tést String", ""};
    strs = NVStrings::create_from_array(sp.data(), sp.size());
    std::vector<NVStrings*> r;
    strs->split_record("s", -1, r);
    EXPECT(r.size() == 5 && r[1] == nullptr);
    if (r.size() == 5) {
      EXPECT(verify_strings(r[0], {"Héllo the", "é"}));
      EXPECT(verify_strings(r[3], {"té", "t String"}));
      EXPECT(verify_strings(r[4], {""}));
    }
    for (auto* p : r) NVStrings::destroy(p);
    r.clear();
    EXPECT(strs->rpartition(" ", r) == 5);
    if (r.size() == 5) {
      EXPECT(verify_strings(r[0], {"Héllo", " ", "thesé"}));
      EXPECT(verify_strings(r[1], {nullptr, nullptr, nullptr}));
      EXPECT(verify_strings(r[4], {"", "", ""}));
    }
    for (auto* p : r) NVStrings::destroy(p);
    // replace_re with several patterns (cpp/tests/replace_multi.cpp:24-58)
    std::vector<const char*> m{"hello there, good friend!", "hi there!", nullptr, "", "!accénted"};
    NVStrings* ms = NVStrings::create_from_array(m.data(), m.size());
    std::vector<const char*> pats{",", "!", "e"};
    const char* one[] = {"_"};
    NVStrings* rp = NVStrings::create_from_array(one, 1);
    got = ms->replace_re(pats, *rp);
    EXPECT(verify_strings(got, {"h_llo th_r__ good fri_nd_", "hi th_r__", nullptr, "", "_accént_d"}));
    NVStrings::destroy(got);
    NVStrings::destroy(rp);
    NVStrings::destroy(ms);
    NVStrings::destroy(strs);
  }
  {  // category remap family and text counters (python/tests/test_category.py:87-247, cpp/tests/test_text.cu:28-103)
    std::vector<const char*> e{"a", "b", "b", "f", "c", "f"};
    NVStrings* strs = NVStrings::create_from_array(e.data(), e.size());
    NVCategory* cat = NVCategory::create_from_strings(*strs);
    int pos[] = {1, 3, 2, 3, 1, 2};
    NVCategory* g = cat->gather_and_remap(pos, 6, false);
    NVStrings* keys = g->get_keys();
    EXPECT(verify_strings(keys, {"b", "c", "f"}));
    int v[6], ev[] = {0, 2, 1, 2, 0, 1};
    g->get_values(v, false);
    for (int i = 0; i < 6; ++i) EXPECT(v[i] == ev[i]);
    NVStrings::destroy(keys);
    NVCategory::destroy(g);
    NVStrings* back = cat->to_strings();
    EXPECT(verify_strings(back, {"a", "b", "b", "f", "c", "f"}));
    NVStrings::destroy(back);
    EXPECT(cat->get_value("c") == 2 && cat->get_value("zz") == -1 && cat->get_value(3u) == 3);
    const char* add[] = {"a", "b", "c", "d"};
    NVStrings* more = NVStrings::create_from_array(add, 4);
    NVCategory* ak = cat->add_keys_and_remap(*more);
    keys = ak->get_keys();
    EXPECT(verify_strings(keys, {"a", "b", "c", "d", "f"}));
    NVStrings::destroy(keys);
    NVCategory::destroy(ak);
    NVStrings::destroy(more);
    bool threw = false;
    int bad[] = {0, 4};
    try {
      cat->gather_strings(bad, 2, false);
    } catch (const std::out_of_range&) {
      threw = true;
    }
    EXPECT(threw);
    NVCategory::destroy(cat);
    NVStrings::destroy(strs);
    std::vector<const char*> t{"the fox jumped over the dog", "the dog chased the cat", "the cat chased the mouse", nullptr, "", "the mouse ate the cheese"};
    strs = NVStrings::create_from_array(t.data(), t.size());
    unsigned int tc[6], etc[] = {6, 5, 5, 0, 0, 5};
    NVText::token_count(*strs, " ", tc, false);
    for (int i = 0; i < 6; ++i) EXPECT(tc[i] == etc[i]);
    NVStrings* ut = NVText::unique_tokens(*strs);
    EXPECT(verify_strings(ut, {"ate", "cat", "chased", "cheese", "dog", "fox", "jumped", "mouse", "over", "the"}));
    NVStrings::destroy(ut);
    const char* q[] = {"cat", "dog"};
    NVStrings* qs = NVStrings::create_from_array(q, 2);
    unsigned int cnt[12], ec[] = {0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
    NVText::tokens_counts(*strs, *qs, " ", cnt, false);
    for (int i = 0; i < 12; ++i) EXPECT(cnt[i] == ec[i]);
    NVStrings::destroy(qs);
    NVStrings::destroy(strs);
  }
  printf(failures ? "%d FAILURES\n" : "all host-API tests passed\n", failures);
  return failures ? 1 : 0;
}
