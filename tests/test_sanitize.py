"""AddressSanitizer + UBSan over the host-only part of the library (N4: csrc/wire.hip has no device code), on the CPU --
GPU sanitizers are not available on the pool.  tests/sanitize/fuzz_wire.cpp drives every decoder with random and
adversarial bytes, round-trips what decodes, and calls the transcript with sizes around the STROBE rate."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_wire_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_wire")
    cmd = [HIPCC, "--offload-host-only", "-x", "hip", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ark_plonk_amd", "csrc"),
           os.path.join(ROOT, "ark_plonk_amd", "csrc", "wire.hip"), "-x", "c++", os.path.join(ROOT, "tests", "sanitize", "fuzz_wire.cpp"),
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "fuzz_wire ok" in run.stdout and "runtime error" not in run.stderr
    shutil.rmtree(tmp_path, ignore_errors=True)
