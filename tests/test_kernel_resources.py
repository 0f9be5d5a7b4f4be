"""Compile-time guard on the registers of the MSM kernels (gfx950 device code, hipcc's own `-Rpass-analysis=kernel-resource-usage`
remarks; no GPU needed).  The accumulation runs at two wavefronts per SIMD because it needs more than 168 and at most 256 registers
and must not spill; `msm_win_finish_q` spilled 162 registers until round 4 and its 512-lane instance must stay clean."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_msm_kernels_registers_and_spills():
    from concurrent.futures import ThreadPoolExecutor

    def remarks(unit):      # the MSM is built as four units per curve (csrc/msm_common.cuh): accumulate / reduce / sort hold the kernels
        src = os.path.join(ROOT, "ark_plonk_amd", "csrc", unit + ".hip")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1000000", "--cuda-device-only",
               "-DZK_CURVE_SEL=0", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
        return subprocess.run(cmd, capture_output=True, text=True, timeout=1500).stderr

    with ThreadPoolExecutor(max_workers=3) as ex:
        err = "\n".join(ex.map(remarks, ("msm_accumulate", "msm_reduce", "msm_sort")))
    kernels, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))

    def find(*parts):
        hits = [v for k, v in kernels.items() if all(p in k for p in parts)]
        assert len(hits) == 1, (parts, [k for k in kernels if parts[0] in k])
        return hits[0]

    acc = find("msm_accumulate_batch")
    assert acc["VGPRs Spill"] == 0 and acc["ScratchSize"] == 0 and 168 < acc["VGPRs"] <= 256 and acc["Occupancy"] == 2, acc
    fin = find("msm_win_finish_q", "Li512E")
    assert fin["VGPRs Spill"] == 0 and fin["VGPRs"] <= 256, fin
    for name in ("psort_scan", "psortw_scatter", "psortw_final", "psort_scatter", "psort_final"):
        k = find(f"{len(name)}{name}E")                     # Itanium mangling: <length><name>E inside the anonymous namespace
        assert k["VGPRs Spill"] == 0 and k["ScratchSize"] == 0, (name, k)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("tool", ["affine_probe.hip", "energy_probe.hip"])
def test_measurement_tools_still_build(tool, tmp_path):
    """The standalone probes behind profiles/r04/r04_notes.md (they include the library's field / group-law headers) compile for gfx950."""
    out = tmp_path / tool.replace(".hip", "")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "ark_plonk_amd", "csrc"), os.path.join(ROOT, "tools", tool),
                        "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and out.exists(), r.stderr[-2000:]
