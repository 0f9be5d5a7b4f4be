#!/bin/bash
# the window width c of the SRS table at one size: bash tools/sweep_window.sh <log_n> <c> <c> ...      (0 = the library's default)
lg=$1; shift
for c in "$@"; do
  timeout -k 10 300 python3 bench.py --log-n $lg --table-window $c --steps 5 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
rf=d['roofline']
print('log_n $lg  c=$c', 'proofs/s', round(d['value'],3), 'ms', round(d['ms_per_step'],2), 'accumulate ms/launch', round(rf['avg_launch_ms'],2), 'adds/scalar', rf['valu']['mixed_adds_per_scalar'], d['commitments_sha256'][:12])" || echo "log_n $lg c=$c FAILED"
done
