import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


CURVE_NAMES = {0: "bls12_381", 1: "bn254"}


@pytest.fixture(scope="session")
def golden():
    out = {}
    for cid, name in CURVE_NAMES.items():
        out[cid] = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    return out


@pytest.fixture(scope="session")
def oracle_cpu():
    from oracle import cpu
    cpu.build()
    cpu.lib()
    return cpu


@pytest.fixture(scope="session")
def ctx():
    """One zk_ctx on GPU 0 for the whole -m gpu session (single process, single card)."""
    import ark_plonk_amd as zk
    c = zk.Context(0)
    yield c
    c.close()


def rand_fr_mont(curve_id, n, seed):
    """n uniformly distributed canonical Fr values as Montgomery limbs (+ the ints)."""
    from oracle import bigint_oracle as bo
    cv = bo.CURVES[curve_id]
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 63, size=(n, 5), dtype=np.uint64)
    vals = []
    for row in raw:
        v = 0
        for k in range(5):
            v |= int(row[k]) << (63 * k)
        vals.append(v % cv.r)
    return vals


# ---- large-size KZG identity helpers: MSM(s, tau^i G) == (sum_i s_i tau^i mod r) G with the O(N) scalar side done by the
# C++ restatement's vectorised Fr ops (a pure-Python big-int loop takes ~1 us per element: minutes at 2^25)
TAU = 0x7A5C0DE


def tau_powers(oracle_cpu, cid, n):
    """(canonical, Montgomery) limbs of tau^i, i < n, by doubling: pw[k:2k] = pw[0:k] * tau^k."""
    from oracle import bigint_oracle as bo
    cv = bo.CURVES[cid]
    pw = np.empty((n, 4), dtype=np.uint64)
    pw[0] = oracle_cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
    k = 1
    while k < n:
        m = min(k, n - k)
        step = oracle_cpu.convert(cid, "fr", True, oracle_cpu.ints_to_limbs([pow(TAU, k, cv.r)], 4))
        pw[k:k + m] = oracle_cpu.fr_op(cid, "mul", pw[:m], np.broadcast_to(step, (m, 4)))
        k *= 2
    return oracle_cpu.convert(cid, "fr", False, pw), pw


def sum_scalar_times_powers(oracle_cpu, cid, scal_raw, pw_mont):
    """sum_i scal_raw[i] * tau^i mod r as a Python int.  Reading the raw limbs as the Montgomery form of v_i = raw_i / R, the
    Montgomery product with mont(tau^i) is mont(v_i tau^i) and the sum's limbs read as an integer are R * sum v_i tau^i
    = sum raw_i tau^i mod r -- no conversion passes."""
    prod = oracle_cpu.fr_op(cid, "mul", scal_raw, pw_mont[: scal_raw.shape[0]])
    while prod.shape[0] > 1:
        h = prod.shape[0] // 2
        s = oracle_cpu.fr_op(cid, "add", prod[:h], prod[h:2 * h])
        prod = s if prod.shape[0] % 2 == 0 else np.concatenate([s, prod[2 * h:]])
    return oracle_cpu.limbs_to_ints(prod)[0]


def srs_from_powers(ctx, cid, pw_canon):
    """tau^i G on the device (the library's fixed-base utility) for canonical scalars pw_canon."""
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    cv = zk.get_curve(cid)
    n = pw_canon.shape[0]
    d = torch.from_numpy(pw_canon.view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d.data_ptr(), n, bases.data_ptr()))
    torch.cuda.synchronize()
    return bases


def assert_is_scalar_times_g(pt, k, cid):
    """pt (G1Affine, Montgomery limbs) == k * G, by the big-int group law."""
    import ark_plonk_amd as zk
    from oracle import bigint_oracle as bo
    cv = bo.CURVES[cid]
    exp = bo.ec_mul(cv, k % cv.r, (cv.gx, cv.gy))
    if exp is None:
        assert pt.infinity
        return
    assert not pt.infinity
    assert zk.curves.fq_from_mont(cid, pt.x.reshape(1, -1))[0] == exp[0]
    assert zk.curves.fq_from_mont(cid, pt.y.reshape(1, -1))[0] == exp[1]
