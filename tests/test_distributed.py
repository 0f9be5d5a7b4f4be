"""N > 1 path: point-sharded MSM = per-rank Jacobian partial -> all-gather -> zk_g1_sum_partials.
CPU: world_size-2 gloo, partials taken from the golden vectors (no device compute).
GPU: two ranks share cuda:0, compute their shard's partial with the HIP MSM, gather with gloo."""
import os
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_cpu(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import ark_plonk_amd as zk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "bls12_381.npz"))
    L = 6
    # rank r contributes the golden MSM result over n = (31, 33)[r] as an affine point lifted to Jacobian (Z = 1)
    n = (31, 33)[rank]
    one = zk.curves.fq_to_mont(0, [1])[0]
    part = np.concatenate([g[f"msm_srs_{n}_out"], one]).astype(np.uint64)
    mine = torch.from_numpy(part.view(np.int64))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    allp = torch.stack(gathered).numpy().view(np.uint64)
    got = zk.sum_partials(allp, 0)
    q.put((rank, got.xy().tolist(), got.infinity))
    dist.destroy_process_group()


def test_gloo_partial_sum_world2():
    import torch.multiprocessing as mp
    from oracle import bigint_oracle as bo
    import ark_plonk_amd as zk
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "bls12_381.npz"))
    cv = bo.BLS12_381
    pts = []
    for n in (31, 33):
        v = zk.curves.fq_from_mont(0, g[f"msm_srs_{n}_out"].reshape(2, 6))
        pts.append((v[0], v[1]))
    exp = bo.ec_add(cv, pts[0], pts[1])
    for rank, xy, inf in res:
        assert not inf
        v = zk.curves.fq_from_mont(0, np.array(xy, dtype=np.uint64).reshape(2, 6))
        assert (v[0], v[1]) == exp, f"rank {rank}"


def _worker_gpu(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import ark_plonk_amd as zk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "bls12_381.npz"))
    n = 1024
    lo, hi = rank * n // world, (rank + 1) * n // world
    ctx = zk.Context(0)
    ck = zk.CommitterKey(g["srs_1024"][lo:hi], 0, ctx)           # this rank's SRS shard
    sc = torch.from_numpy(g["msm_srs_1024_scalars"][lo:hi].view(np.int64)).cuda()
    part = ck.msm_partial(sc, 0)
    mine = torch.from_numpy(part.view(np.int64))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    got = zk.sum_partials(torch.stack(gathered).numpy().view(np.uint64), 0)
    q.put((rank, got.xy().tolist(), got.infinity))
    ck.close()
    ctx.close()
    dist.destroy_process_group()


def _worker_mixed_tables(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import hashlib
    import torch
    import torch.distributed as dist
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log_n = 15
    n = 1 << log_n
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    ks = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ks[:, 0] = torch.arange(3, 3 + 2 * n, 2, device="cuda")
    srs = torch.empty((n, 12), dtype=torch.int64, device="cuda")
    zk._lib.check(zk._lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, srs.data_ptr()))
    lo, hi = rank * n // world, (rank + 1) * n // world
    ck = zk.CommitterKey(srs[lo:hi].contiguous(), 0, ctx).precompute(18 if rank == 0 else 0)   # rank 0: a table without a device form
    sched = ProofSchedule(log_n, ctx, ck, 0, rank=rank, world=world, dist=dist, exchange="winsums")
    pts = sched.run_once()
    dig = hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in pts)).hexdigest()
    q.put((rank, sched.exchange, ck.winsums_geometry() is None, dig))
    ck.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_schedule_takes_the_host_form_when_one_rank_has_no_device_form():
    """Two ranks ask for the window-sum exchange; rank 0's table has 18-bit windows (no device form: what the default is for shards of 2^22
    points and more).  The agreement collective of the schedule runs on BOTH ranks and takes both to the host form -- nobody waits in a
    collective the other skipped -- and the 29 points equal the single rank's."""
    import hashlib
    import torch
    import torch.multiprocessing as mp
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker_mixed_tables, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["host", "host"] and [r[2] for r in res] == [True, False] and res[0][3] == res[1][3]
    log_n = 15
    n = 1 << log_n
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    ks = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ks[:, 0] = torch.arange(3, 3 + 2 * n, 2, device="cuda")
    srs = torch.empty((n, 12), dtype=torch.int64, device="cuda")
    zk._lib.check(zk._lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, srs.data_ptr()))
    ck = zk.CommitterKey(srs, 0, ctx).precompute()
    pts = ProofSchedule(log_n, ctx, ck, 0).run_once()
    assert hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in pts)).hexdigest() == res[0][3]
    ck.close()
    ctx.close()


@pytest.mark.gpu
def test_sharded_msm_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "bls12_381.npz"))
    for rank, xy, inf in res:
        assert not inf and np.array_equal(np.array(xy, dtype=np.uint64), g["msm_srs_1024_out"]), f"rank {rank}"


@pytest.mark.gpu
def test_bench_two_ranks_matches_one_rank():
    """bench.py's N > 1 paths on one card with gloo -- replicas, and point-sharded MSMs + all-gather with
    replicated NTTs: the 29 commitments must equal the single-rank run's, bit for bit."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [os.path.join(ROOT, "bench.py"), "--log-n", "15", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--check"]
    one = subprocess.run([sys.executable] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    def two_ranks(extra):
        # no external torchrun: `bench.py --gpus 2` starts its two ranks itself (both on this box's one card, gloo)
        env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        two = subprocess.run([sys.executable] + common + ["--gpus", "2", "--backend", "gloo"] + extra,
                             capture_output=True, text=True, env=env2, timeout=900)
        assert two.returncode == 0, two.stderr[-3000:]
        return json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][-1])

    # default: replicas are the headline (weak scaling), the sharded-MSM leg rides along and must agree
    d2 = two_ranks([])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak"
    assert d1["commitments_sha256"] == d2["commitments_sha256"]
    assert d2["msm_sharded"].get("commitments_match_replicas") is True, d2["msm_sharded"]
    assert "2^15" in d2["metric"] and "msm_sharded_n22" not in d2        # the 2^22 leg rides along only with the default workload
    # --mode shard: the sharded path as the headline (strong scaling)
    d3 = two_ranks(["--mode", "shard"])
    assert d3["n_gpus"] == 2 and d3["scaling"] == "strong"
    assert d1["commitments_sha256"] == d3["commitments_sha256"]
    # the same sharded by WINDOWS (rank g: the whole SRS, table rows g, g + 2, ...), and with round 3's host-partial exchange
    d4 = two_ranks(["--mode", "shard", "--shard-axis", "windows"])
    assert d4["scaling"] == "strong" and "by windows" in d4["config"]["parallelism"]
    assert d1["commitments_sha256"] == d4["commitments_sha256"]
    d5 = two_ranks(["--mode", "shard", "--shard-axis", "windows", "--host-partials"])
    assert d1["commitments_sha256"] == d5["commitments_sha256"]
    # the two forms of the exchange: host Jacobians (round 3; the default: faster on one card) and window sums on the device (round 5)
    assert "'host'" in d3["config"]["parallelism"] and "'host'" in d4["config"]["parallelism"] and "'host'" in d5["config"]["parallelism"]
    d6 = two_ranks(["--mode", "shard", "--exchange", "winsums"])
    assert "'winsums'" in d6["config"]["parallelism"] and d1["commitments_sha256"] == d6["commitments_sha256"]
    d7 = two_ranks(["--mode", "shard", "--exchange", "winsums", "--shard-axis", "windows"])
    assert "'winsums'" in d7["config"]["parallelism"] and d1["commitments_sha256"] == d7["commitments_sha256"]
    # every N > 1 line says what the backend saw: two processes, ONE card here (a gloo rehearsal), the all-reduced sum of ones
    for d in (d2, d3, d4, d5, d6, d7):
        rk = d["ranks"]
        assert rk["world"] == 2 and rk["sum_check"] == 2 and rk["distinct_devices"] == 1 and rk["shared_card"] is True and len(rk["devices"]) == 2
    assert "ranks" not in d1


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-launches itself as N ranks through torch.distributed.run (one per GPU,
    rendezvous on 127.0.0.1) before touching the GPU; under an external torchrun it does not.  CPU check of the command line."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(8) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main(): N > 1 and no WORLD_SIZE -> self-launch and exit with the children's code, nothing else imported or run
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7


# ---- the exchange step itself (prover_schedule.all_gather_partials): one all_gather of every job's Jacobian partial per group of PC calls
class _StubDist:
    """What the `nccl` branch sees of torch.distributed, single process: records the tensors it is handed and plays `world` ranks
    that all contributed the same partials."""

    def __init__(self, backend, world, with_into_tensor=True):
        self.backend, self.world, self.calls = backend, world, []
        if with_into_tensor:
            self.all_gather_into_tensor = self._into

    def get_backend(self):
        return self.backend

    def _into(self, out, inp):
        self.ptrs = getattr(self, "ptrs", []) + [inp.data_ptr()]
        self.calls.append(("into", out.device, inp.device, out.dtype, inp.dtype, tuple(out.shape), tuple(inp.shape)))
        out.view(self.world, -1).copy_(inp.unsqueeze(0).expand(self.world, -1))

    def all_gather(self, outs, inp):
        self.calls.append(("list", outs[0].device, inp.device, outs[0].dtype, inp.dtype, (len(outs),) + tuple(outs[0].shape), tuple(inp.shape)))
        for o in outs:
            o.copy_(inp)


@pytest.mark.parametrize("into", [True, False])
def test_all_gather_partials_shapes_and_placement_cpu(into):
    """dtype / shape / device of what goes into the collective, on the CPU device (gloo's placement; the nccl branch differs only
    in `device`, exercised on the card by the gpu test below)."""
    import torch
    from ark_plonk_amd.prover_schedule import all_gather_partials
    world, jobs, l3 = 4, 5, 18
    parts = np.arange(jobs * l3, dtype=np.uint64).reshape(jobs, l3) + (1 << 63)       # top bit set: survives the int64 view
    d = _StubDist("gloo", world, with_into_tensor=into)
    got = all_gather_partials(d, parts, world, torch.device("cpu"))
    assert got.shape == (world, jobs, l3) and got.dtype == np.uint64
    for r in range(world):
        assert np.array_equal(got[r], parts)
    kind, odev, idev, odt, idt, oshape, ishape = d.calls[0]
    assert kind == ("into" if into else "list") and odev.type == idev.type == "cpu" and odt == idt == torch.int64
    assert ishape == (jobs * l3,) and oshape == ((world * jobs * l3,) if into else (world, jobs * l3))


def _worker_gather(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ark_plonk_amd.prover_schedule import all_gather_partials
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = (np.arange(3 * 18, dtype=np.uint64).reshape(3, 18) + 1000 * (rank + 1))
    got = all_gather_partials(dist, parts, world, torch.device("cpu"))
    q.put((rank, got.tolist()))
    dist.destroy_process_group()


def test_all_gather_partials_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gather, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(3 * 18, dtype=np.uint64).reshape(3, 18)
    for rank in (0, 1):
        got = np.array(res[rank], dtype=np.uint64)
        assert got.shape == (2, 3, 18)
        assert np.array_equal(got[0], base + 1000) and np.array_equal(got[1], base + 2000)


def _worker_gather_dev(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ark_plonk_amd.prover_schedule import all_gather_partials_dev
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.arange(3 * 32, dtype=torch.int64).reshape(3, 32) + 1000 * (rank + 1)     # 3 jobs x 32 words (a 256-byte partial)
    got = all_gather_partials_dev(dist, mine, world)
    q.put((rank, tuple(got.shape), got.tolist()))
    dist.destroy_process_group()


def test_all_gather_partials_dev_world2_gloo():
    """The device form's exchange step with a real world-2 collective (gloo, CPU tensors standing in for the rank's GPU):
    rank-major (world, jobs x words), every rank sees both ranks' rows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gather_dev, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (sh, v) for r, sh, v in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(3 * 32, dtype=np.int64)
    for rank in (0, 1):
        sh, v = res[rank]
        assert sh == (2, 96)
        assert np.array_equal(np.array(v[0]), base + 1000) and np.array_equal(np.array(v[1]), base + 2000)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["winsums", "host"])
def test_schedule_nccl_branch_places_partials_on_the_rank_gpu(ctx, exchange, monkeypatch):
    """ProofSchedule with world = 2 and a `dist` that reports backend "nccl": the partials of every group of PC calls must enter
    the collective as int64 tensors on the rank's GPU (RCCL cannot take host tensors), one all_gather per group; with both "ranks"
    contributing this rank's shard the result is twice the shard's commitment.
    on_device (the default since round 4): NO host copy between the last reduction kernel and the collective -- the tensor the
    collective sends is the very buffer the library wrote the window sums into (zk_kzg_round_end_winsums_dev), the host-partial entry
    point is never called, and the ranks' sums are added on the device (zk_g1_sum_winsums_dev)."""
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    from ark_plonk_amd.prover_schedule import ProofSchedule
    cv = zk.get_curve(0)
    log_n = 13
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(9)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, bases.data_ptr()))
    ck_full = zk.CommitterKey(bases, cv, ctx).precompute()
    ck_shard = zk.CommitterKey(bases[: n // 2].contiguous(), cv, ctx).precompute()      # rank 0 of 2 owns SRS[0, n/2)
    d = _StubDist("nccl", 2)
    on_device = exchange != "host"
    sched = ProofSchedule(log_n, ctx, ck_shard, cv, rank=0, world=2, dist=d, exchange=exchange)
    assert sched.partials_on_device is on_device and sched.exchange == exchange
    host_calls = []
    real_end_partial = ck_shard.round_end_partial
    monkeypatch.setattr(ck_shard, "round_end_partial", lambda *a, **k: (host_calls.append(1), real_end_partial(*a, **k))[1])
    out = sched.run_once(proof_id=0)
    assert len(out) == 29 and len(d.calls) == 5 and sched.collectives == 5       # five groups of PC calls, one collective each
    words = {"winsums": ck_shard.winsums_dev_words(), "host": 3 * cv.fq_limbs}[exchange]
    assert ck_shard.winsums_dev_words() == 2 * ck_shard.winsums_geometry()[2] * 32 == 4096      # 128 points x 256 B
    for kind, odev, idev, odt, idt, oshape, ishape in d.calls:
        assert odev.type == idev.type == "cuda" and odev.index == idev.index == ctx.device and odt == idt == torch.int64
        assert ishape[0] % words == 0 and oshape == (2 * ishape[0],)
    assert [c[6][0] // words for c in d.calls] == [4, 3, 2, 4, 16]                 # jobs per group
    if on_device:
        assert not host_calls and all(p == sched._pbuf.data_ptr() for p in d.ptrs)
    else:
        assert len(host_calls) == 5
    # rank 0's shard of w_l committed by both stub ranks = 2 * commit(w_l[: n/2]) over SRS[: n/2]
    half = ck_shard.commit(sched.coef[0][: n // 2])
    two = zk.msm.sum_partials(np.stack([ck_shard.commit_batch_partial([sched.coef[0][: n // 2]])[0]] * 2), 0)
    assert out[0] == two and out[0] != half
    ck_full.close()
    ck_shard.close()


def _worker_rccl_world1(port, q):
    """One rank, backend "nccl" (= RCCL on ROCm): the schedule's five exchanges go through a REAL RCCL all_gather of the buffer the
    library wrote (world 1 is all one card allows: RCCL refuses two ranks on one device).  The schedule is told world = 2; the shim
    hands RCCL's output to both rows, i.e. a second rank that owns the same shard."""
    import sys
    sys.path.insert(0, ROOT)
    try:
        import torch
        import torch.distributed as dist
        import ark_plonk_amd as zk
        from ark_plonk_amd import _lib
        from ark_plonk_amd.prover_schedule import ProofSchedule
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            probe = torch.ones(4, dtype=torch.int64, device="cuda")
            dist.all_reduce(probe)                            # RCCL's own bring-up on this box, before any code of this repository
            torch.cuda.synchronize()
        except Exception as e:      # an RCCL that does not come up on a box is not this library's failure
            q.put(("no_rccl", repr(e)))
            return
        q.put(("rccl_up",))

        class TwoRowsOverRccl:
            calls = []

            @staticmethod
            def get_backend():
                return dist.get_backend()

            @classmethod
            def all_gather_into_tensor(cls, out, inp):
                got = torch.empty_like(inp)
                dist.all_gather_into_tensor(got, inp)             # RCCL, on its own stream, ordered after torch's current stream
                cls.calls.append((inp.data_ptr(), inp.numel(), str(inp.device)))
                out.view(2, -1).copy_(got.unsqueeze(0).expand(2, -1))

        cv = zk.get_curve(0)
        ctx = zk.Context(0)
        log_n = 13
        n = 1 << log_n
        g = torch.Generator(device="cuda").manual_seed(9)
        ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        ks[:, 1:] = 0
        bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
        ctx.use_torch_stream()
        _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, bases.data_ptr()))
        ck = zk.CommitterKey(bases[: n // 2].contiguous(), cv, ctx).precompute()
        res = {}
        for exchange in ("winsums", "host"):
            on_device = exchange != "host"
            TwoRowsOverRccl.calls = []
            sched = ProofSchedule(log_n, ctx, ck, cv, rank=0, world=2, dist=TwoRowsOverRccl, exchange=exchange)
            out = sched.run_once(proof_id=0)
            two = zk.msm.sum_partials(np.stack([ck.commit_batch_partial([sched.coef[0][: n // 2]])[0]] * 2), 0)
            res[exchange] = dict(n=len(out), collectives=sched.collectives, calls=len(TwoRowsOverRccl.calls),
                                  devices=sorted({c[2] for c in TwoRowsOverRccl.calls}), first_ok=(out[0] == two),
                                  from_pbuf=all(c[0] == sched._pbuf.data_ptr() for c in TwoRowsOverRccl.calls) if on_device else None,
                                  points=[(pt.infinity, [int(v) for v in pt.x], [int(v) for v in pt.y]) for pt in out])
        ck.close()
        ctx.close()
        dist.destroy_process_group()
        q.put(("ok", res["host"], res["winsums"]))
    except BaseException as e:      # the parent reports it
        import traceback
        q.put(("error", repr(e), traceback.format_exc()[-3000:]))


@pytest.mark.gpu
def test_schedule_exchange_over_real_rccl_world1():
    """The exchange step with RCCL itself in the loop: backend "nccl", one rank on this box's card.  Both forms of the partials (left
    on the device / read back and re-uploaded) give the same 29 points, five collectives each, every tensor RCCL is handed lives on
    cuda:0, and in the device form it is the library's own buffer."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_worker_rccl_world1, args=(_free_port(), q))
    p.start()
    import queue
    try:
        try:
            first = q.get(timeout=240)
        except queue.Empty:
            first = ("no_rccl", "init_process_group('nccl') did not return within 240 s")
        got = q.get(timeout=420) if first[0] == "rccl_up" else first
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    if got[0] == "no_rccl":
        pytest.skip(f"RCCL did not come up on this box: {got[1]}")
    assert got[0] == "ok", got[1:]
    host, ws = got[1], got[2]
    for r in (host, ws):
        assert r["n"] == 29 and r["collectives"] == 5 and r["calls"] == 5 and r["devices"] == ["cuda:0"] and r["first_ok"]
    assert ws["from_pbuf"] is True
    assert host["points"] == ws["points"]


@pytest.mark.gpu
def test_config3_size_two_concurrent_ranks_with_a_collective():
    """BASELINE config 3's size (2^22 points per MSM) as bench.py runs it on N > 1: two ranks CONCURRENTLY on this box's one card,
    every MSM point-sharded (2^21 points and a 4 GiB window table per rank), one all_gather of the Jacobian partials per group
    of PC calls (gloo here, RCCL on a node), NTTs replicated.  The 29 commitments must equal the single-rank run's."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = [os.path.join(ROOT, "bench.py"), "--log-n", "22", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--extra-legs", "off",
              "--streams-leg", "0", "--no-profile"]
    one = subprocess.run([sys.executable] + common, capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.strip().splitlines() if ln.startswith("{")][-1])
    two = subprocess.run([sys.executable] + common + ["--gpus", "2", "--backend", "gloo", "--mode", "shard"],
                         capture_output=True, text=True, env=env, timeout=1200)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and "2^22" in d2["metric"]
    assert d1["commitments_sha256"] == d2["commitments_sha256"]


@pytest.mark.gpu
def test_winsums_form_with_short_and_empty_jobs(ctx):
    """The window-sum form of a round's result (round 5: zk_kzg_round_end_winsums_dev + zk_g1_sum_winsums_dev) with every kind of job:
    table path, a vector too short for it (computed at begin: S_0 = the point, every other sum infinite), the point at infinity
    (all-zero scalars); one "rank" = the blocking batch, two equal "ranks" = twice the commitment as the host form computes it."""
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    cv = zk.get_curve(0)
    n = 1 << 14
    rng = np.random.default_rng(78)
    ks = torch.from_numpy(rng.integers(1, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, bases.data_ptr()))
    ck = zk.CommitterKey(bases, cv, ctx).precompute()
    c_bits, W, VW, VB = ck.winsums_geometry()
    assert VW * VB == 1 << (c_bits - 1) and VW == 64
    polys = [torch.from_numpy(rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64).view(np.int64)).cuda() for m in (n, 100, n - 1)]
    polys.append(torch.zeros((n, 4), dtype=torch.int64, device="cuda"))
    want = ck.commit_batch(polys)
    ww = ck.winsums_dev_words()
    assert ww == 2 * VW * 32                              # 2 VW points of 256 bytes
    buf = torch.full((len(polys), ww), -1, dtype=torch.int64, device="cuda")
    for p in polys:
        ck.commit_begin([p])
    ck.round_reduce_winsums_dev(buf)
    with pytest.raises(RuntimeError):
        ck.round_end(len(polys))                          # reduced towards the device: the host form refuses, the round stays open
    other = torch.zeros_like(buf)
    with pytest.raises(RuntimeError):
        ck.round_end_winsums_dev(other, len(polys))       # ... and so does the device form towards another buffer
    ck.round_end_winsums_dev(buf, len(polys))
    got = ck.sum_winsums_dev(buf.reshape(1, -1), 1, len(polys))
    assert got == want and got[3].infinity
    two = ck.sum_winsums_dev(torch.cat([buf.reshape(1, -1)] * 2), 2, len(polys))
    host2 = zk.sum_partials_batch(np.stack([ck.commit_batch_partial(polys)] * 2), 0)
    assert two == host2
    # three ranks whose sums cancel pairwise where they can: rank rows [P, P, 0] -> 2P again, through another row order
    zero = torch.zeros_like(buf.reshape(1, -1))
    assert ck.sum_winsums_dev(torch.cat([buf.reshape(1, -1), zero, buf.reshape(1, -1)]), 3, len(polys)) == host2
    # the geometry is part of the contract between the ranks: another virtual-window count changes the buffer size
    ctx.set_option("pre_vw", 32)
    try:
        assert ck.winsums_dev_words() == ww // 2 and ck.winsums_geometry()[2] == 32
        buf32 = torch.zeros((2, ww // 2), dtype=torch.int64, device="cuda")
        ck.commit_begin(polys[:2])
        ck.round_end_winsums_dev(buf32, 2)
        assert ck.sum_winsums_dev(buf32.reshape(1, -1), 1, 2) == want[:2]
    finally:
        ctx.set_option("pre_vw", 0)
    ck.close()


@pytest.mark.gpu
def test_device_form_refused_before_anything_is_queued_c18(ctx):
    """a table with window_bits >= 18 finishes its reduction towards the host only.  The device form must say
    ZK_ERR_UNSUPPORTED BEFORE the sort and the merged accumulation of the round's deferred jobs are queued -- otherwise the fall-back
    the header names (zk_kzg_round_end_partial) re-plans jobs that already ran with the long-chunk plan and combines their
    chunk-edge partials wrongly.  Two and three jobs at 2^18 (several rounds of lanes: long chunks differ from short ones)."""
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    cv = zk.get_curve(0)
    n = 1 << 18
    rng = np.random.default_rng(79)
    ks = torch.from_numpy(rng.integers(1, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), n, bases.data_ptr()))
    ck = zk.CommitterKey(bases, cv, ctx).precompute(18)
    assert ck.table_window_bits() == 18 and ck.winsums_geometry() is None and ck.winsums_dev_words() == 0
    polys = [torch.from_numpy(rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64).view(np.int64)).cuda() for m in (n, n - 1, n)]
    want = ck.commit_batch(polys)
    buf = torch.zeros((3, 4096), dtype=torch.int64, device="cuda")
    L = _lib.lib()
    end_dev, red_dev = L.zk_kzg_round_end_winsums_dev, L.zk_kzg_round_reduce_winsums_dev
    for k in (2, 3):
        ck.commit_begin(polys[:k])
        assert red_dev(ctx.handle, buf.data_ptr()) == _lib.ZK_ERR_UNSUPPORTED
        assert end_dev(ctx.handle, k, buf.data_ptr()) == _lib.ZK_ERR_UNSUPPORTED
        assert ck.round_pending() == k                                    # the round is as it was
        parts = ck.round_end_partial(k)                                   # the fall-back the header names
        # a Jacobian triple is not canonical (its Z depends on the order of the additions): compare the points
        assert [zk.sum_partials(parts[j:j + 1], 0) for j in range(k)] == want[:k]
        ck.commit_begin(polys[:k])
        assert end_dev(ctx.handle, k, buf.data_ptr()) == _lib.ZK_ERR_UNSUPPORTED
        assert ck.round_end(k) == want[:k]
    ck.close()


def _run_bench(args, timeout, env_extra=None):
    import json
    import subprocess
    import sys
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(env_extra or {})
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines, time.monotonic() - t0


def test_bench_rehearsal_8_ranks_gloo():
    """`bench.py --gpus 8` as the driver will start it -- eight processes, rendezvous on 127.0.0.1, the `ranks` handshake, barriers, the
    max-over-ranks clock, five all-gathers per proof of the sharded exchange's real shapes (4|3|2|4|16 jobs x 32 KiB) with every
    rank checking every row -- without a GPU and without compute (--rehearse; gloo).  One line, rc 0, bounded wall time."""
    r, lines, wall = _run_bench(["--gpus", "8", "--backend", "gloo", "--rehearse", "--steps", "3", "--warmup", "1"], 240)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and wall < 120
    d = lines[0]
    assert d["rehearsal"] is True and d["value"] is None and d["n_gpus"] == 8 and d["collectives_per_step"] == 5
    rk = d["ranks"]
    assert rk["world"] == 8 and rk["sum_check"] == 8 and rk["backend"] == "gloo" and len(rk["devices"]) == 8
    assert [x.split(":")[0] for x in rk["devices"]] == [str(i) for i in range(8)]
    assert d["bytes_per_rank_per_step"] == 29 * 32768


@pytest.mark.parametrize("where", ["step", "leg"])
def test_bench_rehearsal_killed_rank_ends_the_job(where):
    """One of eight ranks dies between two collectives (os._exit, no clean-up): the job ends non-zero within a bound -- nobody hangs in
    an all_gather.  Killed inside an extra leg (after the headline region was timed), rank 0 still prints the line it has, marked."""
    r, lines, wall = _run_bench(["--gpus", "8", "--backend", "gloo", "--rehearse", "--steps", "3", "--warmup", "1", "--fault-rank", "5",
                                 "--fault-at", where, "--dist-timeout", "60"], 240)
    assert r.returncode != 0 and wall < 90, (r.returncode, wall)
    assert "exitcode: 41" in r.stderr or "exitcode  : 41" in r.stderr
    if where == "step":
        assert lines == []                                     # no value existed yet: nothing may be printed
    else:
        assert len(lines) == 1 and lines[0]["aborted_in_leg"] == "rehearsal_leg" and lines[0]["ranks"]["world"] == 8


def test_bench_rehearsal_rank_out_of_step_inside_a_leg():
    """Nobody dies: one of four ranks stays alive inside the extra leg and never reaches its next collective -- torchrun has nothing to tear
    down and the backend's timeout would end in abort().  Rank 0's leg deadline prints the headline line it already has, marked, and
    ends the job non-zero well before --dist-timeout."""
    r, lines, wall = _run_bench(["--gpus", "4", "--backend", "gloo", "--rehearse", "--steps", "2", "--warmup", "1", "--fault-rank", "2",
                                 "--fault-at", "hang", "--leg-timeout", "8", "--dist-timeout", "120"], 240)
    assert r.returncode != 0 and wall < 90, (r.returncode, wall)
    assert len(lines) == 1 and lines[0]["aborted_in_leg"] == "rehearsal_leg" and "--leg-timeout" in lines[0]["aborted_why"]
    assert lines[0]["ranks"]["world"] == 4 and lines[0]["ms_per_step"] > 0


def test_bench_nccl_refuses_ranks_without_a_card_each():
    """Backend nccl on a host that shows a rank no GPU: exit 2 BEFORE the rendezvous (nobody left waiting).  Ranks that share a card
    are caught by the handshake after it (exit 3 on every rank: `test_ranks_handshake_counts_distinct_devices` checks the count)."""
    r, lines, wall = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], 120,
                                {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2", "MASTER_PORT": str(_free_port()),
                                 "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode == 2 and lines == [] and "one GPU per rank" in r.stderr and wall < 60


def test_ranks_handshake_counts_distinct_devices():
    """The `ranks` object: distinct (host, device) pairs, the all-reduced sum, rank order -- with a stub dist that plays four ranks
    of which two share a card."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class D:
        class ReduceOp:
            SUM = "sum"

        @staticmethod
        def all_gather_object(out, me):
            devs = ["0000:05:00", "0000:15:00", "0000:15:00", "0000:25:00"]
            for r in range(4):
                out[3 - r] = dict(me, rank=r, local_rank=r, device=devs[r], pid=100 + r)      # arrives in any order

        @staticmethod
        def all_reduce(t, op=None):
            t.mul_(4)

    info = bench.ranks_handshake(D, torch, "gloo", 4, 0, 0, None)
    assert info["world"] == 4 and info["sum_check"] == 4 and info["distinct_devices"] == 3 and info["shared_card"] is True
    assert [d.split(":")[0] for d in info["devices"]] == ["0", "1", "2", "3"] and info["hosts"] == 1


@pytest.mark.gpu
def test_bench_gpus_4_one_card():
    """The driver's N > 1 command on this box's ONE card: `bench.py --gpus 4 --backend gloo` (four ranks + this process = five on the
    card, inside the pool's process guard; eight ranks are rehearsed without the GPU above).  Replicas with the msm_sharded leg riding
    along, then the sharded form as the headline on both axes: the 29 commitments equal the single rank's, the line says what the
    backend saw, wall time bounded."""
    common = ["--log-n", "13", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--check"]
    r1, l1, _ = _run_bench(common, 600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = l1[-1]
    r4, l4, wall = _run_bench(common + ["--gpus", "4", "--backend", "gloo"], 900)
    assert r4.returncode == 0, r4.stderr[-3000:]
    assert len(l4) == 1 and wall < 600
    d4 = l4[0]
    assert d4["n_gpus"] == 4 and d4["scaling"] == "weak" and d4["commitments_sha256"] == d1["commitments_sha256"]
    assert d4["ranks"]["world"] == 4 and d4["ranks"]["sum_check"] == 4 and d4["ranks"]["distinct_devices"] == 1
    ms = d4["msm_sharded"]
    assert ms.get("commitments_match_replicas") is True and ms["exchange"] == "host" and ms["collectives_per_proof"] == 5, ms
    mw = d4["msm_sharded_winsums"]          # the same leg with the other form of the exchange: an N > 1 run times both
    assert mw.get("commitments_match_replicas") is True and mw["exchange"] == "winsums" and mw["collectives_per_proof"] == 5, mw
    # wall seconds per leg (DESIGN.md 6 extrapolates the driver's 8-GPU run from the same fields of a 2^20 rehearsal): the budget of
    # this rehearsal -- four ranks time-slicing ONE card at 2^13 -- is a minute per leg and the sum is what the run took
    ls = d4["leg_s"]
    assert set(ls) >= {"headline", "msm_sharded", "msm_sharded_winsums", "total"} and all(v < 60 for k, v in ls.items() if k != "total"), ls
    assert abs(ls["total"] - sum(v for k, v in ls.items() if k != "total")) < 5 and ls["total"] < wall
    hp = d4["ranks"]["host_pool"]
    assert hp["local_world"] == 4 and hp["workers_per_ctx"] == max(0, min(15, hp["host_cores"] // 4 - 1)), hp
    for axis, form in (("points", "winsums"), ("windows", "host")):
        r, l, _ = _run_bench(common + ["--gpus", "4", "--backend", "gloo", "--mode", "shard", "--shard-axis", axis, "--exchange", form], 900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert l[-1]["scaling"] == "strong" and l[-1]["commitments_sha256"] == d1["commitments_sha256"], axis


@pytest.mark.gpu
def test_bench_killed_rank_inside_the_sharded_leg():
    """Three ranks on the card, rank 1 dies at the start of the msm_sharded leg (after the replicas' headline region): the job ends
    non-zero within a bound and rank 0's line -- the headline it had already measured -- is printed with `aborted_in_leg`."""
    r, lines, wall = _run_bench(["--log-n", "13", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--gpus", "3", "--backend", "gloo",
                                 "--fault-rank", "1", "--dist-timeout", "60"], 600)
    assert r.returncode != 0 and wall < 300, (r.returncode, wall)
    assert len(lines) == 1 and lines[0]["aborted_in_leg"] == "msm_sharded" and lines[0]["value"] > 0 and lines[0]["n_gpus"] == 3
