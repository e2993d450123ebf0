#!/usr/bin/env python3
"""Dev probe (GPU box): wall time of API variants that may route to the row-wise kernels."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custrings_amd import _lib, nvstrings, nvtext, nvcategory
L = _lib.lib; _lib.ensure_init(0)
def synth(kind, rows, param=0):
    out = C.c_void_p(); _lib.check(L.cs_synth_column(kind, 0, rows, 20240607, param, None, C.byref(out))); return nvstrings.nvstrings(out.value)
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
rows = 20_000_000
c3 = synth(3, rows)
resb = torch.empty(rows, dtype=torch.uint8, device="cuda"); resi = torch.empty(rows, dtype=torch.int32, device="cuda")
IP = r"\d+\.\d+\.\d+\.\d+"
cases = [
 ("replace('.', '_') literal", lambda: c3.replace(".", "_", regex=False)),
 ("replace('/', '//') literal", lambda: c3.replace("/", "//", regex=False)),
 ("replace('GET', 'G') literal", lambda: c3.replace("GET", "G", regex=False)),
 ("replace_re(IPv4,'<IP>')", lambda: c3.replace(IP, "<IP>")),
 ("replace_re(IPv4,'<IP>', n=1)", lambda: c3.replace(IP, "<IP>", 1)),
 ("match(IPv4)", lambda: c3.match(IP, devptr=resb.data_ptr())),
 ("contains_re(IPv4)", lambda: c3.contains(IP, devptr=resb.data_ptr())),
 ("contains_re('\\\\bGET\\\\b')", lambda: c3.contains(r"\bGET\b", devptr=resb.data_ptr())),
 ("contains_re('[45]0[0-9] ')", lambda: c3.contains(r"[45]0[0-9] ", devptr=resb.data_ptr())),
 ("count_re('\\\\d+')", lambda: c3.count(r"\d+", devptr=resi.data_ptr())),
 ("contains_re(GET|POST|PUT|DELETE|HEAD)", lambda: c3.contains(r"GET|POST|PUT|DELETE|HEAD", devptr=resb.data_ptr())),
 ("contains_re('\\w+@\\w+\\.\\w+')", lambda: c3.contains(r"\w+@\w+\.\w+", devptr=resb.data_ptr())),
 ("contains_re('\\d{4}-\\d{2}-\\d{2}')", lambda: c3.contains(r"\d{4}-\d{2}-\d{2}", devptr=resb.data_ptr())),
 ("contains_re('/[a-z]+/[a-z]+')", lambda: c3.contains(r"/[a-z]+/[a-z]+", devptr=resb.data_ptr())),
 ("contains_re('^(GET|POST) /a')", lambda: c3.contains(r"^(GET|POST) /a", devptr=resb.data_ptr())),
 ("contains_re(' (4|5)\\d\\d( |$)')", lambda: c3.contains(r" (4|5)\d\d( |$)", devptr=resb.data_ptr())),
 ("replace_re('\\s+', ' ')", lambda: c3.replace(r"\s+", " ")),
 ("replace_re('[aeiou]', '')", lambda: c3.replace(r"[aeiou]", "")),
 ("replace_re('(GET|POST) ', 'M ')", lambda: c3.replace(r"(GET|POST) ", "M ")),
 ("replace_re('\\b\\d{1,3}(\\.\\d{1,3}){3}\\b','<IP>')", lambda: c3.replace(r"\b\d{1,3}(\.\d{1,3}){3}\b", "<IP>")),
 ("split(' ')", lambda: c3.split(" ")),
 ("split(' ', 2)", lambda: c3.split(" ", 2)),
 ("split()", lambda: c3.split()),
 ("split(None, 2)", lambda: c3.split(None, 2)),
 ("split('. ')  (2-byte delimiter)", lambda: c3.split(". ")),
 ("strip()", lambda: c3.strip()),
 ("lstrip('GETPOS ')", lambda: c3.lstrip("GETPOS ")),
 ("rstrip('0123456789 ')", lambda: c3.rstrip("0123456789 ")),
 ("lower()", lambda: c3.lower()),
 ("find('200')", lambda: c3.find("200", devptr=resi.data_ptr())),
 ("find('200', 10, 60)", lambda: c3.find("200", 10, 60, devptr=resi.data_ptr())),
 ("contains('POST', regex=False)", lambda: c3.contains("POST", regex=False, devptr=resb.data_ptr())),
 ("tokenize()", lambda: nvtext.tokenize(c3)),
 ("tokenize(' ./')", lambda: nvtext.tokenize(c3, " ./")),
 ("byte_count", lambda: c3.byte_count(resi.data_ptr(), bdevmem=True)),
 ("null_count", lambda: c3.null_count()),
]
print("rows = %d (C3 shape, %.2f GB chars)" % (rows, L.cs_column_nbytes(c3.m_cptr) / 1e9))
for name, fn in cases:
    try:
        print("%-36s %9.3f ms" % (name, t(fn)), flush=True)
    except Exception as e:
        print("%-36s ERROR %s" % (name, e), flush=True)
tok = nvtext.tokenize(c3)
print("%-36s %9.3f ms" % ("ngrams(tokens,3,'_')", t(lambda: nvtext.ngrams(tok, 3, "_"))))
c4a, c4b = synth(4, 10_000_000, 100000), synth(4, 10_000_000, 50000)
ca, cb = nvcategory.from_strings(c4a), nvcategory.from_strings(c4b)
print("%-36s %9.3f ms" % ("category merge (2 x 10M rows)", t(lambda: nvcategory.from_categories([ca, cb]))))
print("%-36s %9.3f ms" % ("category from 2 columns", t(lambda: nvcategory.from_strings(c4a, c4b))))
