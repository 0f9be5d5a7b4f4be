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
