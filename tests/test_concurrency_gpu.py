"""Thread-safety of the boundary (SURVEY.md 8b "Threading": the ABI must be safe for concurrent calls on one
ctx; rayon-parallel callers exist in the ark ecosystem) and independence of several contexts on one GPU."""
import threading

import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from oracle import bigint_oracle as bo

pytestmark = pytest.mark.gpu


def _srs(ctx, cv, n, torch):
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = 5 + 3 * np.arange(n, dtype=np.uint64)
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cv.curve_id, torch.from_numpy(ks.view(np.int64)).cuda().data_ptr(), n,
                                                     bases.data_ptr()))
    return bases


def test_threads_sharing_one_context(ctx, oracle_cpu):
    """Four host threads interleave NTTs and commitments on ONE zk_ctx (host-buffer entry points, which share the
    ctx's staging buffers): every result equals the serial one."""
    import torch
    cid, cv, log_n = 0, bo.CURVES[0], 12
    n = 1 << log_n
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    srs = _srs(ctx, cv, n, torch)
    srs_h = srs.cpu().numpy().view(np.uint64)
    ck = zk.CommitterKey(srs_h, cid, ctx)
    rng = np.random.default_rng(3)
    polys = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(4)]
    exp_ntt = [oracle_cpu.ntt(cid, 1, log_n, p) for p in polys]
    exp_cm = [oracle_cpu.kzg_commit(cid, srs_h, p)[0] for p in polys]
    errs = []

    def worker(k):
        try:
            for _ in range(6):
                got = dom.ifft(polys[k])                     # host arrays -> zk_ntt (io_a / io_b staging)
                assert np.array_equal(got, exp_ntt[k]), "ifft"
                cm = ck.commit(polys[k])                     # zk_kzg_commit (scalar staging + MSM buffers)
                assert np.array_equal(cm.xy(), exp_cm[k]), "commit"
        except Exception as e:   # surfaced below
            errs.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ck.close()
    assert not errs, errs


def test_two_contexts_on_one_gpu_in_parallel(ctx):
    """Two zk_ctx (own stream each, driven from two threads) give the results of the shared test context."""
    import torch
    cid, cv, log_n = 0, bo.CURVES[0], 14
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(5)
    poly = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    srs = _srs(ctx, cv, n, torch)
    ck0 = zk.CommitterKey(srs, cid, ctx).precompute()
    ref_cm = ck0.commit_batch([poly, poly])[0]
    ref_ev = zk.Radix2EvaluationDomain.new(n, cid, ctx).coset_fft(poly).cpu()
    ck0.close()
    torch.cuda.synchronize()
    out, errs = {}, []

    def worker(k):
        try:
            c = zk.Context(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                st.wait_stream(torch.cuda.default_stream())
                ck = zk.CommitterKey(srs, cid, c).precompute()
                dom = zk.Radix2EvaluationDomain.new(n, cid, c)
                for _ in range(5):
                    ev = dom.coset_fft(poly)
                    cm = ck.commit_batch([poly, poly])
                st.synchronize()
                out[k] = (ev.cpu(), cm[0], cm[1])
                ck.close()
            c.close()
        except Exception as e:
            errs.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        ev, a, b = out[k]
        assert torch.equal(ev, ref_ev) and a == ref_cm and b == ref_cm
