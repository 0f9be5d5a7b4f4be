"""Device memory of one proof of the headline schedule at a size: python tools/mem_probe.py [log_n]   (prints used GiB after the SRS, the table,
the schedule's own vectors and one proof; at 2^25 the deferred 16-job round does not fit beside them -- profiles/r05_notes.md section 8)"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import ark_plonk_amd as zk
from ark_plonk_amd.prover_schedule import ProofSchedule
from bench import build_srs
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = 1 << log_n
ctx = zk.Context(0); ctx.use_torch_stream()
cv = zk.get_curve(0)
def mem(tag):
    f, t = torch.cuda.mem_get_info()
    print(f"{tag}: used {(t - f) / 2**30:.1f} GiB of {t / 2**30:.1f}; torch allocated {torch.cuda.memory_allocated() / 2**30:.1f} reserved {torch.cuda.memory_reserved() / 2**30:.1f}", flush=True)
mem("start")
srs = build_srs(ctx, cv, n, 0, n, torch)
mem("srs built")
ck = zk.CommitterKey(srs, cv, ctx).precompute()
del srs
torch.cuda.empty_cache()
mem("table built")
s = ProofSchedule(log_n, ctx, ck, cv)
mem("schedule inputs")
try:
    s.run_once()
    mem("after one proof")
except Exception as e:
    print("failed:", e)
    mem("at failure")
