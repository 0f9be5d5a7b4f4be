# Kernel traces of BASELINE configs 3 (2^22, one GPU) and 4 (BN254, 2^18); run on the GPU box, then tools/summarize_rocprof.py on the merged gpurun_out/prof_r04_cfgs/{n22,bn18} -> profiles/r04/r04_bench_n22.{md,csv}, r04_bench_bn254_n18.{md,csv}
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r04_cfgs; mkdir -p $out
B22="bench.py --log-n 22 --steps 2 --warmup 1 --extra-legs off --streams-leg 0 --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/n22 -o t -- python3 $B22 > $out/n22.log 2>&1 || { tail -5 $out/n22.log; exit 1; }
python3 tools/summarize_rocprof.py $out/n22 $out/r04_bench_n22 "rocprofv3 --kernel-trace --stats -- python3 $B22" > $out/n22_summary.txt
BBN="bench.py --curve bn254 --log-n 18 --steps 5 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bn18 -o t -- python3 $BBN > $out/bn18.log 2>&1 || { tail -5 $out/bn18.log; exit 1; }
python3 tools/summarize_rocprof.py $out/bn18 $out/r04_bench_bn254_n18 "rocprofv3 --kernel-trace --stats -- python3 $BBN" > $out/bn18_summary.txt
tail -3 $out/n22.log | cut -c1-300; tail -3 $out/bn18.log | cut -c1-300
