"""Sanitizers over the HOST side of the library, on the CPU -- GPU sanitizers are not available on the pool.
* AddressSanitizer + UBSan over csrc/wire.hip (N4: no device code): tests/sanitize/fuzz_wire.cpp drives every decoder with random
  and adversarial bytes, round-trips what decodes, and calls the transcript with sizes around the STROBE rate.
* ASan + UBSan, and ThreadSanitizer, over the host side of the WHOLE library: every translation unit compiled with
  `hipcc --offload-host-only` (kernels become host stubs) and linked against tests/sanitize/fake_hip.cpp instead of the HIP runtime
  (device memory = zeroed host memory, launches do nothing); tests/sanitize/host_stress.cpp then runs the SRS registry, the commitment
  cache, blocking batches, deferred rounds, the device form of the partials and the host pool from several threads -- the pattern of
  tests/test_concurrency_gpu.py -- plus the host-only point arithmetic (zk_g1_sum_partials*) on real points."""
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


def _lib_units():
    import sys
    sys.path.insert(0, ROOT)
    from ark_plonk_amd import build as zk_build
    return [(obj, src, defs) for obj, src, defs, _ in zk_build.jobs()]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_side_of_the_library_under_sanitizers(tmp_path, san):
    from concurrent.futures import ThreadPoolExecutor
    clangxx = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clangxx):
        clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    csrc = os.path.join(ROOT, "ark_plonk_amd", "csrc")
    flags = ["--offload-host-only", "-x", "hip", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-sanitize-recover=undefined",
             "-Wno-option-ignored", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-I", csrc]
    cmds, objs = [], []
    for obj, src, defs in _lib_units():
        o = str(tmp_path / obj)
        objs.append(o)
        cmds.append([HIPCC] + flags + [d for d in defs if d.startswith("-D")] + ["-c", os.path.join(csrc, src), "-o", o])
    o = str(tmp_path / "fake_hip.o")
    objs.append(o)
    cmds.append([HIPCC] + flags + ["-c", os.path.join(ROOT, "tests", "sanitize", "fake_hip.cpp"), "-o", o])
    o = str(tmp_path / "host_stress.o")
    objs.append(o)
    cmds.append([clangxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-I", os.path.join(ROOT, "include"), "-c",
                 os.path.join(ROOT, "tests", "sanitize", "host_stress.cpp"), "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr[-3000:]

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, cmds))
    exe = str(tmp_path / "host_stress")
    # every HIP object references its (absent) device image by a per-file symbol that only __hipRegisterFatBinary -- a no-op here -- receives
    run([clangxx, f"-fsanitize={san}", "-Wl,--unresolved-symbols=ignore-all", "-o", exe] + objs + ["-lpthread"])
    env = {**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
           "TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"}
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "host_stress ok" in r.stdout
    assert "runtime error" not in out and "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out, out[-4000:]
    shutil.rmtree(tmp_path, ignore_errors=True)
