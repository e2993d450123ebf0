"""SURVEY.md section 5 (sanitizers): the CPU builds of the checker (oracle) and of the product's host-compiled row logic
(tests/rowemu: row ops, regex compiler, tagged-DFA builder and executor, the lean scans) run a slice of the parity suite
under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on this pool; the device code
shares these headers.  The instrumented libraries are loaded in a child interpreter with libasan preloaded."""
import os
import subprocess
import sys

import cpulibs

ROOT = cpulibs.ROOT


def _runtime(name):
    out = subprocess.run(["g++", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_and_row_emulation_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    assert asan, "g++ has no libasan.so"
    for d in ("oracle", os.path.join("tests", "rowemu")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, d), "asan"], check=True)
    env = dict(os.environ, CS_SANITIZE="1", LD_PRELOAD=asan + ((":" + ubsan) if ubsan else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the golden vectors through the oracle, the host emulation against the oracle (fuzzed rows, generated programs, the unit
    # decomposition and the lean scans), the compiler against the reference's programs
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
           os.path.join(ROOT, "tests", "test_rowemu_parity.py"), "-k", "not slow"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
