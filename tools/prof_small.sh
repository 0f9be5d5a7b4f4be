set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s2
mkdir -p $out
B="--steps 10 --warmup 3 --extra-legs off --streams-leg 0 --no-cpu-baseline"
timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline > $out/warm.log 2>&1
for cfg in "bn254 18" "bls12_381 18" "bls12_381 16"; do
  set -- $cfg
  tag=$1_$2
  timeout -k 10 200 python3 bench.py --curve $1 --log-n $2 $B > $out/bench_$tag.json 2> $out/bench_$tag.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -o t -- python3 bench.py --curve $1 --log-n $2 $B > $out/trace_$tag.log 2>&1 || exit 1
  python3 tools/summarize_rocprof.py $out/trace_$tag $out/prof_$tag "rocprofv3 --kernel-trace --stats -- python3 bench.py --curve $1 --log-n $2 $B" > /dev/null
  python3 tools/trace_idle.py $out/trace_$tag >> $out/prof_$tag.md
  rm -rf $out/trace_$tag
  cut -c1-300 $out/bench_$tag.json
done
