"""CPU-only: the per-row STL math the HIP kernels run, exercised under AddressSanitizer + UBSan (GPU sanitizers are not
available on the pool, so the host build of csrc/stl_core.hpp is what gets sanitized)."""
import os
import subprocess

from conftest import ROOT


def test_stl_core_is_clean_under_asan_and_ubsan(tmp_path):
    src = os.path.join(ROOT, "tests", "hostsim", "sanitize_main.cpp")
    exe = str(tmp_path / "sanitize_main")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-Wno-unknown-pragmas", src, "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "row evaluations clean" in p.stdout
