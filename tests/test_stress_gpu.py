"""Reduced-iteration runs of the randomised cross-checks in tests/stress/ (HIP path vs. the CPU restatement, limb for limb):
random sizes, lengths, batch shapes, scalar distributions, entry points.  The full-length runs are logged under profiles/."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "stress", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stress_ntt_short(ctx, oracle_cpu):
    assert _load("stress_ntt").run(budget=12.0, seed=20, ctx=ctx, max_log_n=18) >= 20


def test_stress_msm_short(ctx, oracle_cpu):
    assert _load("stress_msm").run(budget=20.0, seed=10, ctx=ctx, max_log_n=16) >= 4


def test_stress_kzg_short(ctx, oracle_cpu):
    assert _load("stress_kzg").run(budget=12.0, seed=30, ctx=ctx, max_len=1 << 16) >= 10


def test_stress_rounds_short(ctx, oracle_cpu):
    assert _load("stress_rounds").run(budget=12.0, seed=40, ctx=ctx, max_len=1 << 15) >= 20


def test_stress_prover_short(ctx, oracle_cpu):
    assert _load("stress_prover").run(budget=12.0, seed=50, ctx=ctx, max_log_n=9) >= 5


def test_stress_shards_short(ctx, oracle_cpu):
    assert _load("stress_shards").run(budget=20.0, seed=60, ctx=ctx, max_log_n=15) >= 4


def test_stress_residency_short(ctx, oracle_cpu):
    assert _load("stress_residency").run(budget=15.0, seed=70, ctx=ctx, max_log_n=13) >= 20


def test_stress_residency_with_cache_verify(ctx, oracle_cpu):
    """the same randomised session with option "cache_verify": every hit of the residency and commitment caches is checked against the
    caller's bytes / a recomputation (include/ark_plonk_amd.h: what the caches guarantee) -- no mismatch"""
    assert _load("stress_residency").run(budget=10.0, seed=71, ctx=ctx, max_log_n=13, verify=True) >= 10
