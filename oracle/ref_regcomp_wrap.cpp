// TEST INFRASTRUCTURE (oracle side) -- not part of the product.
//
// Thin C wrapper around the REAL reference regex compiler.  It is compiled
// together with /root/reference/cpp/src/regex/regcomp.cpp *where that file
// lies* (see oracle/Makefile, target _ref); nothing from the reference is
// copied into this repository and the resulting library only ever lives in
// oracle/_ref/ (git-ignored).  regcomp.cpp is pure host C++ and needs no
// CUDA/RMM/Thrust, so it builds with plain g++.
//
// The wrapper serialises the reference's Reprog through its public accessors
// into the same int32 blob layout the product uses
// (custrings_amd/csrc/regex_program.h), so tests can diff the two compilers
// word for word and can feed the reference-compiled program to the oracle VM.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "regcomp.h"  // -I/root/reference/cpp/src/regex

extern "C" {

// pattern: packed-UTF-8 chars (one uint32 per char), 0-terminated.
// Returns number of int32 words written to a malloc'd buffer (*out).
int ref_regcomp_blob(const uint32_t* pattern, int32_t** out) {
  Reprog* prog = Reprog::create_from(reinterpret_cast<const char32_t*>(pattern));
  std::vector<int32_t> b(8, 0);
  b[0] = 0x58525343;
  b[1] = prog->get_start_inst();
  b[2] = prog->groups_count();
  b[3] = prog->inst_count();
  b[4] = prog->starts_count();
  b[5] = prog->classes_count();
  const Reinst* insts = prog->insts_data();
  for (int i = 0; i < prog->inst_count(); ++i) {
    b.push_back(insts[i].type);
    b.push_back(insts[i].u1.right_id);
    b.push_back(insts[i].u2.left_id);
    b.push_back(0);  // pad4 is uninitialised in the reference
  }
  const int* starts = prog->starts_data();
  for (int i = 0; i < prog->starts_count(); ++i) b.push_back(starts[i]);
  int32_t off = 0;
  for (int i = 0; i < prog->classes_count(); ++i) {
    b.push_back(off);
    off += 1 + (int32_t)prog->class_at(i).chrs.size();
  }
  b.push_back(off);
  b[6] = off;
  for (int i = 0; i < prog->classes_count(); ++i) {
    Reclass& c = prog->class_at(i);
    b.push_back(c.builtins);
    for (char32_t ch : c.chrs) b.push_back((int32_t)ch);
  }
  delete prog;
  *out = (int32_t*)malloc(b.size() * sizeof(int32_t));
  memcpy(*out, b.data(), b.size() * sizeof(int32_t));
  return (int)b.size();
}

void ref_free(void* p) { free(p); }

}  // extern "C"
