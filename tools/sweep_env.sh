#!/bin/bash
# tuning sweep over one option of the library (zk_ctx_set_option; bench.py --option): bash tools/sweep_env.sh chunk_l 48 64 86 ...
var=$1; shift
for v in "$@"; do
  python bench.py --option $var=$v --steps 10 --warmup 3 --extra-legs off --streams-leg 0 --no-cpu-baseline --check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$var=$v', round(d['value'],3), round(d['ms_per_step'],2), 'msm', round(d['msm_ms_per_proof'],2), 'acc', round(d['roofline']['avg_launch_ms'],4), d['commitments_sha256'][:12])"
done
