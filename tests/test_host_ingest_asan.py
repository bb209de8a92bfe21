"""Sanitizer tier (-m "not gpu"): the HOST C++ of the product -- the two FASTA readers
(mmap, threads, AVX paths, untrusted input; reference idelucs/utils.py:26-51,224-260), check_sequence, the packer -- and the C
oracle are built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU code objects only; the GPU code never is) and driven
over every fixture at 1 / 3 / 8 threads plus a seeded fuzz loop.  The drivers (tests/asan/ingest_driver.cpp,
oracle/asan_driver.c) also cross-check the two readers against each other, so the run fails on a report OR a disagreement."""
import glob
import os
import shutil
import subprocess

import pytest

from conftest import DATA, ROOT

CSRC = os.path.join(ROOT, "idelucs_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           LSAN_OPTIONS="suppressions=" + os.path.join(ROOT, "tests", "asan", "lsan.supp"))


def _sanitizers_usable():
    """g++ can build AND run a trivial -fsanitize=address,undefined program (the runtimes are separate packages)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write("int main() { return 0; }\n")
        r = subprocess.run(["g++", "-fsanitize=address,undefined", src, "-o", exe], capture_output=True)
        return r.returncode == 0 and subprocess.run([exe], capture_output=True).returncode == 0


def _build(directory, target):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host compiler")
    rocm = os.environ.get("ROCM", "/opt/rocm")
    if directory == CSRC and not (os.path.isdir(os.path.join(rocm, "include", "hip")) and glob.glob(os.path.join(rocm, "lib", "libamdhip64.so*"))):
        pytest.skip("no ROCm headers / libamdhip64 to build the host side against")     # (a CPU-only CI box skips, not fails)
    if not _sanitizers_usable():
        pytest.skip("the AddressSanitizer / UBSan runtimes are not installed")
    r = subprocess.run(["make", "-C", directory, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(directory, target)
    assert os.path.exists(exe)
    return exe


def test_host_readers_are_clean_under_asan_and_ubsan(tmp_path):
    exe = _build(CSRC, "host_ingest_asan")
    files = sorted(glob.glob(os.path.join(DATA, "*.fas")))
    assert any(f.endswith("empty.fas") for f in files) and len(files) >= 6
    for seed in ("1", "20261004"):
        r = subprocess.run([exe, str(tmp_path), "300", seed] + files, capture_output=True, text=True, env=ENV, timeout=600)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr[-4000:]
        counts = [int(x) for x in r.stdout.replace(",", " ").split() if x.isdigit()]
        assert len(counts) == 5 and counts[0] > 100 and counts[1] > 10 and counts[2] > 20 and counts[3] > 10, r.stdout   # every outcome exercised


def test_empty_file_does_not_pass_a_null_pointer_to_memchr(tmp_path):
    """The UB the round-3 review found with UBSan (host_ingest.cpp: memchr(buf + id_b, '\\t', 0) with buf == NULL on an empty
    file): the driver's first fuzz case IS the empty file; `-fno-sanitize-recover=undefined` makes any such report fatal."""
    exe = _build(CSRC, "host_ingest_asan")
    r = subprocess.run([exe, str(tmp_path), "1", "0", os.path.join(DATA, "empty.fas")], capture_output=True, text=True, env=ENV, timeout=120)
    assert r.returncode == 0 and "runtime error" not in r.stderr, r.stderr[-2000:]


def test_oracle_is_clean_under_asan_and_ubsan():
    exe = _build(os.path.join(ROOT, "oracle"), "oracle_asan")
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr[-4000:]
    assert "cases clean" in r.stdout
