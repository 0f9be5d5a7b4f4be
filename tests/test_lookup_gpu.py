"""Round 2 of the prover on the device (prover.rs:228-317): compressed query column and `MultiSet::combine_split` through the C
ABI vs the restatement in oracle/bigint_oracle.py, which tests/test_oracle.py pins on the reference's own known-answer vector
(lookup/multiset.rs:335-392)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import lookup  # noqa: E402
from ark_plonk_amd.curves import fr_from_mont, fr_to_mont  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402

pytestmark = pytest.mark.gpu


def dev(cid, ints):
    import torch
    a = fr_to_mont(cid, ints) if len(ints) else np.zeros((0, 4), dtype=np.uint64)
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 4)).cuda()


def back(cid, t):
    return fr_from_mont(cid, t.cpu().numpy().view(np.uint64)) if t.shape[0] else []


@pytest.mark.parametrize("cid", [0, 1])
def test_combine_split_reference_vector(cid, ctx):
    """lookup/multiset.rs:335-392 `test_combine_split`, and the example of the doc comment (:125-130)."""
    h1, h2 = lookup.combine_split(dev(cid, [0, 1, 2, 3, 4, 5, 6]), dev(cid, [3, 6, 0, 5, 4, 3, 2, 0, 0, 1, 2]), cid, ctx)
    assert back(cid, h1) == [0, 0, 1, 2, 2, 3, 4, 5, 6] and back(cid, h2) == [0, 0, 1, 2, 3, 3, 4, 5, 6]
    h1, h2 = lookup.combine_split(dev(cid, [2, 4, 1, 3]), dev(cid, [2, 3, 3, 2]), cid, ctx)
    assert back(cid, h1) == [2, 2, 1, 3] and back(cid, h2) == [2, 4, 3, 3]


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("n_t,n_f,distinct", [(1, 0, 1), (1, 5, 1), (7, 7, 3), (1000, 1000, 37), (4099, 8191, 4099), (70001, 65536, 30000)])
def test_combine_split_vs_oracle(cid, n_t, n_f, distinct, ctx):
    """Repeated table rows (a bucket's name is the FIRST index of its value), one value that most queries hit (the reference's
    dummy row), odd and even bucket sizes, an odd total (the halves differ by one), across workgroup and scan-block edges."""
    cv = bo.CURVES[cid]
    rng = np.random.default_rng(1000 * n_t + n_f + cid)
    pool = bo.seeded_scalars(cv, 0xF000 + n_t, distinct)
    t = [pool[k] for k in rng.integers(0, distinct, n_t)]
    t[0] = pool[0]
    pick = rng.integers(0, n_t, n_f)
    dummy = rng.random(n_f) < 0.6
    f = [t[0] if dummy[i] else t[pick[i]] for i in range(n_f)]
    want1, want2 = bo.combine_split(t, f)
    h1, h2 = lookup.combine_split(dev(cid, t), dev(cid, f), cid, ctx)
    assert back(cid, h1) == want1 and back(cid, h2) == want2
    assert 0 <= len(want1) - len(want2) <= 1


def test_combine_split_element_not_indexed(ctx):
    cv = bo.CURVES[0]
    t = bo.seeded_scalars(cv, 0xF100, 500)
    f = [t[3], t[4], (t[5] + 1) % cv.r if (t[5] + 1) % cv.r not in t else 12345, t[6]]
    with pytest.raises(lookup.ElementNotIndexed):
        lookup.combine_split(dev(0, t), dev(0, f), 0, ctx)
    h1, h2 = lookup.combine_split(dev(0, []), dev(0, []), 0, ctx)          # two empty multisets: two empty halves
    assert h1.shape[0] == 0 and h2.shape[0] == 0
    with pytest.raises(lookup.ElementNotIndexed):
        lookup.combine_split(dev(0, []), dev(0, [1]), 0, ctx)


@pytest.mark.parametrize("cid", [0, 1])
def test_round2_vs_oracle(cid, ctx):
    """prover.rs:228-317 end to end at n = 512: compressed table, query column with a q_lookup shorter than n (zero-padded by
    the reference), h_1 and h_2 -- every row."""
    cv = bo.CURVES[cid]
    n = 512
    rng = np.random.default_rng(77 + cid)
    rows = 40                                                        # distinct table rows, repeated to fill the table (padding)
    base = [bo.seeded_scalars(cv, 0xF200 + k, rows) for k in range(4)]
    idx = list(range(rows)) + [0] * (n - rows)
    table_cols = [[base[k][j] for j in idx] for k in range(4)]
    q_len = 300
    q = [int(v) for v in rng.integers(0, 2, q_len)]
    wires = [bo.seeded_scalars(cv, 0xF210 + k, n) for k in range(4)]
    for i in range(q_len):                                           # lookup rows hold a table row
        if q[i]:
            j = int(rng.integers(0, rows))
            for k in range(4):
                wires[k][i] = base[k][j]
    zeta = bo.seeded_scalars(cv, 0xF220, 1)[0]
    want_t, want_f, want_h1, want_h2 = bo.lookup_round2(cv, n, table_cols, q, wires, zeta)
    zm = fr_to_mont(cid, [zeta])[0]
    t = lookup.compress_table([dev(cid, c) for c in table_cols], zm, cid, ctx)
    f = lookup.compress_query(dev(cid, q), [dev(cid, w) for w in wires], zm, t, curve=cid, ctx=ctx)
    h1, h2 = lookup.combine_split(t, f, cid, ctx)
    assert back(cid, t) == want_t and back(cid, f) == want_f and back(cid, h1) == want_h1 and back(cid, h2) == want_h2
    assert h1.shape[0] == n and h2.shape[0] == n


def test_combine_split_full_size(ctx):
    """n = 2^20 table rows and 2^20 queries (BASELINE config 2), 60 % of them the dummy row: against the restatement (a Python
    dict over 2^21 integers), plus the multiset identity h_1 + h_2 = t + f as a size-independent check."""
    import torch
    cid, n = 0, 1 << 20
    g = torch.Generator(device="cuda").manual_seed(5)
    distinct = 1 << 18
    pool = torch.randint(0, 1 << 62, (distinct, 4), dtype=torch.int64, device="cuda", generator=g)
    pool[:, 3] &= (1 << 60) - 1
    t = pool[torch.randint(0, distinct, (n,), device="cuda", generator=g)].contiguous()
    pick = torch.randint(0, n, (n,), device="cuda", generator=g)
    pick[torch.rand(n, device="cuda", generator=g) < 0.6] = 0
    f = t[pick].contiguous()
    h1, h2 = lookup.combine_split(t, f, cid, ctx)
    assert h1.shape[0] == n and h2.shape[0] == n
    key = lambda a: [bytes(r) for r in a.cpu().numpy().view(np.uint8).reshape(-1, 32)]  # noqa: E731
    want1, want2 = bo.combine_split(key(t), key(f))
    assert key(h1) == want1 and key(h2) == want2
    mix = lambda a: int((a.view(torch.int64) * torch.tensor([3, 5, 7, 11], device="cuda")).sum(dim=1).sum().item())  # noqa: E731  (wrapping sums)
    assert (mix(h1) + mix(h2) - mix(t) - mix(f)) % (1 << 64) == 0
