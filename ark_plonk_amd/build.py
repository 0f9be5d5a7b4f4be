"""Build the gfx950 shared library (hipcc, in-tree).  `python -m ark_plonk_amd.build [--force]`.

The heavy kernels are compiled as separate objects -- one per (curve, NTT radix exponent) and four MSM
units per curve (sort / accumulate / reduce / host plan: csrc/msm_common.cuh) -- so a full build
parallelises over the host cores and an edit rebuilds only what it touches.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libark_plonk_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -pragma-unroll-threshold: the NTT passes keep 8 field elements per lane in registers and every loop over
# them must fully unroll (a rolled loop indexes the array at run time and sends it to scratch memory)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-mllvm", "-pragma-unroll-threshold=1000000", "-Wpass-failed"]

FIELD_HDRS = ["field.cuh", "fieldu.cuh", "fields.cuh", "curve_params.h", "zk_common.h"]
HOST_HDRS = FIELD_HDRS + ["ec.cuh", "ecu.cuh", "ctx.h", "../../include/ark_plonk_amd.h"]
MSM_UNITS = ("msm_accumulate", "msm_reduce", "msm_sort", "msm_plan")      # heaviest first


def jobs():
    """(object name, source, extra defines, header deps)"""
    out = [
        ("api.o", "api.hip", [], HOST_HDRS),
        ("hostio.o", "hostio.hip", [], HOST_HDRS),
        ("wire.o", "wire.hip", [], HOST_HDRS),
        ("kzg.o", "kzg.hip", [], HOST_HDRS + ["fr_io.cuh"]),
        ("lookup.o", "lookup.hip", [], HOST_HDRS + ["fr_io.cuh"]),
        ("grand_product.o", "grand_product.hip", [], HOST_HDRS),
        ("quotient.o", "quotient.hip", [], HOST_HDRS),
        ("quadtest.o", "quadtest.hip", [], HOST_HDRS + ["ecq.cuh"]),
        ("ntt.o", "ntt.hip", [], HOST_HDRS + ["ntt_pass.cuh"]),
        ("ntt_pass_table.o", "ntt_pass_table.hip", [], []),
        ("msm_dispatch.o", "msm_dispatch.hip", [], HOST_HDRS),
    ]
    for c in (0, 1):
        # ARK_PLONK_AMD_MSM_FLAGS: extra compiler flags for the MSM objects only (scheduler experiments: tools/ab_bench.sh)
        for unit in MSM_UNITS:
            out.append((f"{unit}_c{c}.o", f"{unit}.hip", [f"-DZK_CURVE_SEL={c}"] + os.environ.get("ARK_PLONK_AMD_MSM_FLAGS", "").split(),
                        HOST_HDRS + ["msm_common.cuh"] + (["ecq.cuh"] if unit == "msm_reduce" else [])))
        for s in range(3, 10):
            out.append((f"ntt_pass_c{c}_s{s}.o", "ntt_pass_inst.hip", [f"-DZK_CURVE_SEL={c}", f"-DZK_NTT_S={s}"],
                        FIELD_HDRS + ["ntt_pass.cuh"]))
    return out


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, workers: int | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    todo = []
    objs = []
    for obj, src, defs, hdrs in jobs():
        o = os.path.join(OBJ, obj)
        objs.append(o)
        deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in hdrs]
        if force or _stale(o, deps):
            todo.append([HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", o])
    # heaviest first (large S, MSM) so the tail of the build is short
    todo.sort(key=lambda c: (0 if "/msm_" in " ".join(c[-3:]) else 1, -int(next((d.split("=")[1] for d in c if d.startswith("-DZK_NTT_S=")), 0))))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stderr[-4000:])
        return r.stderr

    if todo:
        n = workers or min(len(todo), max(1, (os.cpu_count() or 4)))
        with ThreadPoolExecutor(max_workers=n) as ex:
            for err in ex.map(run, todo):
                if verbose and err.strip():
                    print(err, file=sys.stderr)
    if force or todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd[:6]), "...", flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
