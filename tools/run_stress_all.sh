#!/bin/bash
# Every randomised cross-check of tests/stress/ on the GPU box, one after the other, with a seed of the caller's choice:
#   tools/run_stress_all.sh <seconds each> <seed> [MAX_LOG_N for the MSM / shard runs]      logs -> gpurun_out/stress_<seed>/
set -u
ulimit -c 0
secs=${1:-120}; seed=${2:-7}; export MAX_LOG_N=${3:-19}; export SEED=$seed
out=gpurun_out/stress_$seed; mkdir -p "$out"
rc=0
for s in ntt msm kzg rounds prover shards residency; do
  case $s in msm|shards|residency) args="$secs";; *) args="$secs $seed";; esac
  timeout -k 10 $((secs + 240)) python3 tests/stress/stress_$s.py $args > "$out/$s.log" 2>&1 || { rc=1; echo "FAILED: $s"; tail -5 "$out/$s.log"; break; }
  grep "stress ok" "$out/$s.log"
done
exit $rc
