// Compiled-regex program shared by the host compiler, the HIP executor and the
// test/oracle tooling.
//
// The instruction set and its numeric opcodes are the reference's `Reinst`
// contract (/root/reference/cpp/src/regex/regcomp.h:25-65): the executor's
// thread-priority semantics are defined on this instruction stream, so the
// compiler must emit it instruction-for-instruction (checked against the real
// reference compiler in tests/test_regex_compile.py).
//
// Flat "blob" layout (int32 words), used on the wire (C-ABI, LDS staging):
//   [0] magic 'CSRX'   [1] start_inst   [2] num_groups   [3] n_insts
//   [4] n_starts (incl. the -1 terminator)   [5] n_classes
//   [6] class_words (total words in the class-data section)   [7] reserved
//   insts   : n_insts  x {type, u1, u2, 0}
//   starts  : n_starts x int
//   cls_off : (n_classes + 1) x int   word offsets into class data
//   cls_data: per class {builtins, lo0, hi0, lo1, hi1, ...}
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace csrx {

enum Op : int32_t {
  OP_CHAR = 0177,
  OP_RBRA = 0201,
  OP_LBRA = 0202,
  OP_OR = 0204,
  OP_ANY = 0300,
  OP_ANYNL = 0301,
  OP_BOL = 0303,
  OP_EOL = 0304,
  OP_CCLASS = 0305,
  OP_NCCLASS = 0306,
  OP_BOW = 0307,
  OP_NBOW = 0310,
  OP_END = 0377,
};

// builtin class bits (regcomp.cpp:51-56)
enum Builtin : int32_t { BI_w = 1, BI_s = 2, BI_d = 4, BI_W = 8, BI_S = 16, BI_D = 32 };

constexpr int32_t kBlobMagic = 0x58525343;  // "CSRX"
constexpr int kBlobHeaderWords = 8;

struct Inst {
  int32_t type;
  int32_t u1;  // char | class id | group id | OR right (preferred) branch
  int32_t u2;  // next | OR left branch
};

struct CharClass {
  int32_t builtins = 0;
  std::vector<uint32_t> ranges;  // lo,hi pairs of packed-UTF-8 chars
};

struct Program {
  std::vector<Inst> insts;
  std::vector<CharClass> classes;
  std::vector<int32_t> starts;  // -1 terminated
  int32_t start_inst = 0;
  int32_t num_groups = 0;

  std::vector<int32_t> to_blob() const;
  // blob + executor "extras" (layout: regex_vm.h): per-class ASCII bitmaps, the
  // \w ASCII bitmap and the first-character prefilter.  `flags` is the 64 KiB
  // unicode flag table.
  std::vector<int32_t> to_device_image(const uint8_t* flags) const;
};

// Packs a NUL-terminated UTF-8 pattern into one uint32 per character holding
// the raw 1-4 bytes big-endian (NVStringsImpl.cu:49-66 `to_char32`).
std::vector<uint32_t> pack_utf8(const char* s);

// Compiles a packed pattern. Never throws on malformed syntax (the reference
// does not report syntax errors either, regcomp.cpp:193-197).
Program compile(const uint32_t* pattern);
inline Program compile(const char* utf8) {
  auto p = pack_utf8(utf8);
  return compile(p.data());
}

}  // namespace csrx
