#!/bin/bash
# usage: run.sh VARIANT WAVES
cd /tmp/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/ark_plonk_amd/csrc -DVARIANT=$1 -DWAVES=$2 --cuda-device-only --no-gpu-bundle-output -c acc_exp.hip -o acc_$1_$2.co 2>&1 | grep -v warning | head -5
/opt/rocm/lib/llvm/bin/llvm-readelf --notes acc_$1_$2.co | grep -E "vgpr_count|vgpr_spill|private_segment_fixed" | paste - - -
