# tuning hooks of the shared-bucket reduction / accumulation plan at the small configurations (A/B inside one process per size)
set -u
out=gpurun_out/s2; mkdir -p $out
for cfg in "bn254 18" "bls12_381 16" "bls12_381 18"; do
  set -- $cfg
  timeout -k 10 400 python3 tools/ab_proof.py --curve $1 --log-n $2 --pairs 8 --proofs 4 \
    "pre_vw=0" "pre_vw=32" "pre_vw=128" "pre_vw=256" "pre_logg=2" "pre_logg=3" "pre_logg=4" "combine_sg=2" "combine_sg=4" "chunk_l=64" "chunk_l=128" "chunk_l=256" "long_rounds=2" "msm_merge=0" \
    > $out/sweep_$1_$2.txt 2>&1 || exit 1
  cat $out/sweep_$1_$2.txt
done
