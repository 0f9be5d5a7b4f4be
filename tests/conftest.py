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
