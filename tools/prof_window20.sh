#!/bin/bash
# kernel times of the headline schedule with a 20-bit table window (13 rows, 2^19 buckets: the wide reduction) at 2^22 and 2^20
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5/prof_c20; mkdir -p $out
for lg in 22 20; do
  B="bench.py --log-n $lg --table-window ${WINDOW:-20} --steps 3 --warmup 1 --extra-legs off --streams-leg 0 --no-cpu-baseline"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/n$lg -o t -- python3 $B > $out/line_$lg.json 2> $out/err_$lg.txt || { tail -5 $out/err_$lg.txt; exit 1; }
  python3 tools/summarize_rocprof.py $out/n$lg $out/stats_n$lg "rocprofv3 --kernel-trace --stats -- python3 $B" > $out/summary_$lg.txt
  find $out/n$lg -type f -name "*.csv" -size +2M -delete
  head -40 $out/stats_n$lg.md
done
