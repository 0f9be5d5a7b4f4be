#!/bin/bash
# A/B of library builds in ONE GPU session (boxes differ by several per cent, so only alternating runs on one box compare):
#   tools/ab_bench.sh out_dir libA.so libB.so [...]      -> bench.py headline-only runs, A B A B, key numbers per run
# Variant libraries are built with `python -m ark_plonk_amd.build` on a modified tree and copied to tools/bin/ (not tracked).
set -u
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    ARK_PLONK_AMD_LIB="$lib" timeout -k 10 240 python bench.py --steps 8 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline \
      > "$out/${name}_$rep.json" 2> "$out/${name}_$rep.err"
    python - "$out/${name}_$rep.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %7.3f proofs/s  step %7.2f ms  acc %6.4f ms  msm %6.2f  ntt %5.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"],
      d["roofline"]["avg_launch_ms"], d["msm_ms_per_proof"], d["ntt_ms_per_proof"], d.get("commitments_sha256", "")[:12]))
PY
  done
done
