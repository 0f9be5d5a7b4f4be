"""CPU suite: the C-ABI library loads, exports every symbol include/ark_plonk_amd.h declares, and the
host-only entry points (domain constants, error paths, partial-sum combine) agree with the oracle.
No device compute is attempted here."""
import ctypes
import os
import re

import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from oracle import bigint_oracle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ark_plonk_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ark_plonk_amd.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SYMBOLS) == syms


def test_strerror_and_build_info():
    L = _lib.lib()
    assert L.zk_strerror(0) == b"ok"
    assert b"two-adicity" in L.zk_strerror(_lib.ZK_ERR_DOMAIN_TOO_LARGE)
    assert b"gfx950" in L.zk_build_info()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


@pytest.mark.parametrize("cid", [0, 1])
def test_domain_new_matches_golden(cid, golden):
    g = golden[cid]
    cv = bo.CURVES[cid]
    for log_n in (0, 1, 5, 10, 20, cv.two_adicity):
        num = (1 << log_n) if log_n == 0 else (1 << (log_n - 1)) + 1  # next_power_of_two rounding
        d = zk.Radix2EvaluationDomain.new(num, cid)
        assert d is not None and d.size() == 1 << log_n and d.log_size_of_group() == log_n
        assert np.array_equal(d.group_gen(), g[f"group_gen_{log_n}"])
        assert np.array_equal(d.group_gen_inv(), g[f"group_gen_inv_{log_n}"])
        assert np.array_equal(d.size_inv(), g[f"size_inv_{log_n}"])
        assert np.array_equal(d.generator(), g["generator"])
        assert np.array_equal(d.generator_inv(), g["generator_inv"])
    # EvaluationDomain::new returns None past the field's two-adicity (error.rs:14-21)
    assert zk.Radix2EvaluationDomain.new((1 << cv.two_adicity) + 1, cid) is None


def test_bad_arguments_are_codes_not_crashes():
    L = _lib.lib()
    info = _lib.DomainInfo()
    assert L.zk_domain_new(7, 8, ctypes.byref(info)) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_domain_new(0, 8, None) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_ntt(None, 0, 0, 4, None, 0, None) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_msm_g1(None, 0, None, None, None, 0, None, None) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_ctx_create(0, None) == _lib.ZK_ERR_BAD_ARG
    L.zk_ctx_destroy(None)
    L.zk_srs_free(None)
    assert L.zk_srs_len(None) == 0


@pytest.mark.parametrize("cid", [0, 1])
def test_sum_partials_host(cid, golden):
    """zk_g1_sum_partials (the multi-GPU combine) on Jacobian triples built from golden affine points."""
    g = golden[cid]
    cv = bo.CURVES[cid]
    L = cv.fq_limbs
    G = (cv.gx, cv.gy)
    pts = [bo.ec_mul(cv, k, G) for k in (3, 5, 11)]
    parts = []
    for i, p in enumerate(pts):
        z = 1 + 7 * i  # non-trivial Z: (x z^2, y z^3, z)
        parts += [p[0] * z * z % cv.q, p[1] * z * z * z % cv.q, z]
    parts += [1, 1, 0]  # an infinity partial
    arr = zk.curves.fq_to_mont(cid, parts).reshape(-1, 3 * L)
    got = zk.sum_partials(arr, cid)
    exp = bo.ec_mul(cv, 19, G)
    assert not got.infinity
    assert zk.curves.fq_from_mont(cid, got.x.reshape(1, L))[0] == exp[0]
    assert zk.curves.fq_from_mont(cid, got.y.reshape(1, L))[0] == exp[1]
    # P + (-P) = infinity -> (0, 1) + flag, like GroupAffine::zero()
    p = pts[0]
    arr2 = zk.curves.fq_to_mont(cid, [p[0], p[1], 1, p[0], cv.q - p[1], 1]).reshape(2, 3 * L)
    got2 = zk.sum_partials(arr2, cid)
    assert got2.infinity and not got2.x.any()
    assert zk.curves.fq_from_mont(cid, got2.y.reshape(1, L))[0] == 1
    # doubling branch: P + P
    arr3 = zk.curves.fq_to_mont(cid, [p[0], p[1], 1, p[0], p[1], 1]).reshape(2, 3 * L)
    got3 = zk.sum_partials(arr3, cid)
    exp3 = bo.ec_mul(cv, 6, G)
    assert zk.curves.fq_from_mont(cid, got3.x.reshape(1, L))[0] == exp3[0]


@pytest.mark.parametrize("cid", [0, 1])
def test_sum_partials_batch_host(cid):
    """zk_g1_sum_partials_batch: [rank][job][3L] partials of one prover round -> one point per job (threaded host tail)."""
    cv = bo.CURVES[cid]
    L = cv.fq_limbs
    G = (cv.gx, cv.gy)
    ranks, jobs = 3, 20           # > 16 jobs: exercises the bounded thread groups
    flat, exp = [], []
    ks = [[1 + 5 * r + 17 * j for j in range(jobs)] for r in range(ranks)]
    ks[1][4] = 0                  # an infinity partial
    for r in range(ranks):
        for j in range(jobs):
            if ks[r][j] == 0:
                flat += [1, 1, 0]
                continue
            p = bo.ec_mul(cv, ks[r][j], G)
            z = 2 + r + j
            flat += [p[0] * z * z % cv.q, p[1] * z * z * z % cv.q, z]
    arr = zk.curves.fq_to_mont(cid, flat).reshape(ranks, jobs, 3 * L)
    got = zk.sum_partials_batch(arr, cid)
    assert len(got) == jobs
    for j in range(jobs):
        e = bo.ec_mul(cv, sum(ks[r][j] for r in range(ranks)), G)
        assert not got[j].infinity
        assert zk.curves.fq_from_mont(cid, got[j].x.reshape(1, L))[0] == e[0]
        assert zk.curves.fq_from_mont(cid, got[j].y.reshape(1, L))[0] == e[1]
        assert got[j] == zk.sum_partials(arr[:, j, :], cid)


def test_cpp_host_header_compiles():
    """host/ark_plonk_amd.hpp (the C++ mirror of EvaluationDomain / VariableBaseMSM / KZG commit) is valid C++17."""
    import subprocess
    src = '#include "host/ark_plonk_amd.hpp"\nint main() { return sizeof(zk::G1Affine) > 0 ? 0 : 1; }\n'
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-x", "c++", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_null_handles_are_refused_by_every_entry_point():
    """Never `abort` / fault across the boundary (SURVEY.md 8b "Errors"): each of the ABI's functions that takes a ctx, an SRS or a
    transcript handle first returns a negative code when that handle is NULL and every other argument is zero -- in a child process,
    so a fault would fail this test with the function's name instead of ending the run."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_null_probe.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"crashed in {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else '?'} (exit {r.returncode})"
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(res) >= 70
    empty_is_ok = {"zk_g1_sum_winsums_dev"}              # n_jobs = 0: nothing to do, documented as ZK_OK
    for name, rc in res.items():
        assert rc < 0 or name in empty_is_ok, (name, rc)
        if name not in empty_is_ok:
            assert rc == _lib.ZK_ERR_BAD_ARG, (name, rc)


def test_committed_counter_files_name_the_current_kernel_build():
    """bench.py quotes profiles/pmc_*.json (`roofline.traffic`, `roofline_ntt.traffic` / `.issue`): both must carry the commit of the
    kernel sources they were collected from, and -- in a git checkout -- that commit is the last one that touched csrc/."""
    import json
    import subprocess
    a = json.load(open(os.path.join(ROOT, "profiles", "pmc_msm_accumulate.json")))
    n = json.load(open(os.path.join(ROOT, "profiles", "pmc_ntt.json")))
    assert a["commit"] == n["commit"] and a["hbm_bytes_per_launch"] > 0 and n["hbm_bytes_per_proof"] > n["alg_bytes_per_proof_n20"]
    assert 0.5 < n["issue"]["valu_busy_per_simd_weighted"] < 1.2 and len(n["issue"]["kernels"]) >= 3
    r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "ark_plonk_amd/csrc"], capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip():
        assert r.stdout.strip().startswith(a["commit"]) or a["commit"].startswith(r.stdout.strip()), (a["commit"], r.stdout.strip())


def test_curve_constants_and_host_scalar_mul():
    """ark_plonk_amd/curves.py (product code, independent of oracle/): the G1 generators lie on their curves and the pure-Python
    double-and-add bench.py uses for its KZG identity checks agrees with the oracle's group law."""
    from ark_plonk_amd import curves
    from oracle import bigint_oracle as bo
    for cid in (0, 1):
        cv, o = curves.get_curve(cid), bo.CURVES[cid]
        assert (cv.gy * cv.gy - cv.gx ** 3 - cv.b) % cv.q == 0 and (cv.gx, cv.gy, cv.r, cv.q) == (o.gx, o.gy, o.r, o.q)
        for k in (0, 1, 2, 3, 0xDEADBEEF12345678, cv.r - 1, cv.r, cv.r + 5):
            assert curves.g1_mul(cv, k) == bo.ec_mul(o, k % o.r, (o.gx, o.gy)), (cid, k)
