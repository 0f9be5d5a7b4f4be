#!/bin/bash
# tuning sweep of the table path's reduction geometry (virtual windows, buckets per segment): bash tools/sweep_vw.sh "64 2" "64 3" ...
for combo in "$@"; do
  set -- $combo
  python bench.py --option pre_vw=$1 --option pre_logg=$2 --steps 10 --warmup 3 --extra-legs off --streams-leg 0 --no-cpu-baseline --check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('VW=$1 logG=$2', round(d['value'],3), round(d['ms_per_step'],2), 'msm', round(d['msm_ms_per_proof'],2), d['commitments_sha256'][:12])"
done
