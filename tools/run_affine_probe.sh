set -e
ulimit -c 0
P=tools/bin/affine_probe
O=gpurun_out/affine_probe.txt
: > $O
timeout -k 10 120 $P check 16384 64 20 1 >> $O 2>&1
for cfg in "xyzz 131072 600 24" "xyzz 131072 300 24" "level0 131072 300 24" "level0 196608 200 24" "level0 262144 150 24" "level0 131072 150 24" "level0 131072 75 24" "level1 131072 300 24" "level1 131072 150 24" "level1 262144 75 24" "inv 131072 1 24" "inv 262144 1 24"; do
  timeout -k 10 120 $P $cfg 5 >> $O 2>&1
done
cat $O
