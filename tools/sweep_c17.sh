set -u
out=gpurun_out/r3k; mkdir -p $out
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
b() { # label, extra env string, bench args...
  label=$1; envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --steps 6 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline "$@" > $out/$label.json 2> $out/$label.err
  python - "$out/$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-26s %7.3f proofs/s  step %8.2f ms  acc %7.4f  msm %7.2f  ntt %6.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["msm_ms_per_proof"], d["ntt_ms_per_proof"], d.get("commitments_sha256", "")[:10]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
b c17_vw64_g3 "ZK_PRE_VW=64 ZK_PRE_LOGG=3" --table-window 17
b c17_vw64_g4 "ZK_PRE_VW=64 ZK_PRE_LOGG=4" --table-window 17
b c17_vw128_g3 "ZK_PRE_VW=128 ZK_PRE_LOGG=3" --table-window 17
b c17_vw128_g2 "ZK_PRE_VW=128 ZK_PRE_LOGG=2" --table-window 17
b c17_vw32_g4 "ZK_PRE_VW=32 ZK_PRE_LOGG=4" --table-window 17
b c16_base "ZK_X=0" --table-window 16
b n18_c16 "ZK_X=0" --table-window 16 --log-n 18
b n18_c17 "ZK_X=0" --table-window 17 --log-n 18
b n19_c16 "ZK_X=0" --table-window 16 --log-n 19
b n19_c17 "ZK_X=0" --table-window 17 --log-n 19
b n22_c16 "ZK_X=0" --table-window 16 --log-n 22 --steps 3
b n22_c17 "ZK_X=0" --table-window 17 --log-n 22 --steps 3
b bn18_c16 "ZK_X=0" --table-window 16 --log-n 18 --curve bn254
b bn18_c17 "ZK_X=0" --table-window 17 --log-n 18 --curve bn254
