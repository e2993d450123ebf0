"""Relink check of the drop-in boundary: a translation unit compiled against
the REFERENCE's own headers (/root/reference/cpp/include) that calls the members this build covers
must find every symbol it needs in libNVStrings.so / libNVCategory.so / libNVText.so -- same mangled
names, so consumers of the reference relink without recompiling.  Runs only where the reference tree
is present (this container); nothing from it is copied: the headers are included where they lie."""
import os
import subprocess
import tempfile

import pytest

import cpulibs

ROOT = cpulibs.ROOT
REF_INC = "/root/reference/cpp/include"

CALLER = r"""
#include <vector>
#include <utility>
#include "NVStrings.h"
#include "NVCategory.h"
#include "NVText.h"
void calls(NVStrings* s, NVCategory* c, std::vector<NVStrings*>& v, std::vector<NVCategory*>& cv, std::vector<const char*>& pats,
           std::pair<const char*, size_t>* ix, const char** arr, int* ip, unsigned* up, bool* bp, char* cp, unsigned char* ucp, char** list) {
  NVStrings::create_from_array(arr, 1); NVStrings::create_from_index(ix, 1); NVStrings::create_from_offsets(cp, 1, ip);
  NVStrings::create_from_strings(v); NVStrings::destroy(s);
  s->size(); s->memsize(); s->create_index(ix); s->create_offsets(cp, ip); s->set_null_bitarray(ucp); s->copy(); s->to_host(list, 0, 1);
  s->sublist(0, 1); s->gather(ip, 1); s->gather(bp); s->scatter(*s, ip); s->scatter("x", ip, 1); s->remove_strings(ip, 1);
  s->sort(); s->order(NVStrings::name, true, up); s->len(ip); s->byte_count(ip);
  s->cat(s, ":"); s->cat(v, ":"); s->join();
  s->split_record(":", 1, v); s->rsplit_record(":", 1, v); s->split_record(1, v); s->rsplit_record(1, v);
  s->split(":", 1, v); s->rsplit(":", 1, v); s->split(1, v); s->rsplit(1, v); s->partition(":", v); s->rpartition(":", v);
  s->extract("a", v); s->extract_record("a", v); s->findall("a", v); s->findall_record("a", v);
  s->replace("a", "b"); s->replace_re("a", "b"); s->replace_re(pats, *s); s->replace_with_backrefs("a", "b");
  s->lstrip(" "); s->strip(" "); s->rstrip(" "); s->lower(); s->upper();
  s->find("a", 0, -1, ip); s->contains("a", bp); s->contains_re("a", bp); s->match("a", bp); s->count_re("a", ip);
  // the rest of find.cu (NVStrings.h:849-934)
  s->compare("a", ip); s->rfind("a", 0, -1, ip); s->find_from("a", ip, ip, ip); s->find_multiple(*s, ip); s->match_strings(*s, bp);
  s->startswith("a", bp); s->endswith("a", bp);
  NVCategory::create_from_array(arr, 1); NVCategory::create_from_index(ix, 1); NVCategory::create_from_offsets(cp, 1, ip);
  NVCategory::create_from_strings(*s); NVCategory::create_from_strings(v); NVCategory::create_from_categories(cv); NVCategory::destroy(c);
  c->get_type_name(); reinterpret_cast<base_category_type*>(c)->get_type_name();
  c->size(); c->keys_size(); c->has_nulls(); c->copy(); c->get_keys(); c->get_value(0u); c->get_value("a"); c->get_values(ip); c->values_cptr();
  c->get_indexes_for(0u, ip); c->get_indexes_for("a", ip); c->add_strings(*s); c->remove_strings(*s); c->add_keys_and_remap(*s);
  c->remove_keys_and_remap(*s); c->set_keys_and_remap(*s); c->remove_unused_keys_and_remap(); c->merge_category(*c); c->merge_and_remap(*c);
  c->to_strings(); c->gather_strings(ip, 1); c->gather_and_remap(ip, 1); c->gather(ip, 1);
  NVText::tokenize(*s); NVText::tokenize(*s, *s); NVText::unique_tokens(*s); NVText::token_count(*s, " ", up); NVText::tokens_counts(*s, *s, " ", up);
  NVText::replace_tokens(*s, *s, *s); NVText::normalize_spaces(*s); NVText::create_ngrams(*s, 2, "_");
}
"""


def _symbols(args):
    out = subprocess.run(["nm"] + args, capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference tree is not on this box")
def test_reference_compiled_callers_relink_against_our_libraries():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "custrings_amd", "host"), "libs"], check=True)
    with tempfile.TemporaryDirectory() as d:
        src, obj = os.path.join(d, "caller.cpp"), os.path.join(d, "caller.o")
        open(src, "w").write(CALLER)
        subprocess.run(["g++", "-std=c++14", "-c", "-I", REF_INC, src, "-o", obj], check=True)
        wanted = {s for s in _symbols(["-u", obj]) if "NVStrings" in s or "NVCategory" in s or "NVText" in s}
    have = set()
    for lib in ("libNVStrings.so", "libNVCategory.so", "libNVText.so"):
        have |= _symbols(["-D", "--defined-only", os.path.join(ROOT, "custrings_amd", lib)])
    assert len(wanted) > 97
    assert not (wanted - have), sorted(wanted - have)


BASE_USER = r"""
#include <cstdio>
#include <cstring>
#include "nvstrings/NVCategory.h"
// what python/cpp/numeric_category.cpp:217-236 does with a category handle: cast to the base, dispatch on the name
int main() {
  NVCategory* c = NVCategory::adopt(nullptr);  // (an empty instance: no device needed)
  base_category_type* b = reinterpret_cast<base_category_type*>(c);
  const char* name = b->get_type_name();
  std::printf("%s\n", name);
  const bool ok = std::strcmp(name, "custring") == 0 && *reinterpret_cast<void**>(c) != nullptr;
  NVCategory::destroy(c);
  return ok ? 0 : 1;
}
"""


def test_category_is_reachable_through_base_category_type():
    """base_category.h:18-23 / NVCategory.h:48: the object's first word is a vtable pointer and get_type_name()
    dispatched through the base says "custring" (NVCategory.cu:581)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "custrings_amd", "host"), "libs"], check=True)
    libdir = os.path.join(ROOT, "custrings_amd")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "base_user.cpp"), os.path.join(d, "base_user")
        open(src, "w").write(BASE_USER)
        subprocess.run(["g++", "-std=c++14", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-lNVCategory", "-lNVStrings",
                        "-Wl,-rpath," + libdir], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "custring", (out.returncode, out.stdout, out.stderr)
    syms = _symbols(["-D", "--defined-only", os.path.join(libdir, "libNVCategory.so")])
    assert "_ZTV10NVCategory" in syms and "_ZN10NVCategory13get_type_nameEv" in syms
