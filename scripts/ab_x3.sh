#!/bin/bash
# Ablation probe of the bf16x3 GEMM K loop (not part of the product): builds gemm_x3.hip + gemm.hip stand-alone with X3_ABL = each listed
# value into ab/libx3_<v>.so; run `python scripts/ab_x3.py` on the GPU box afterwards.
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
[ -f ab/gemm.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o ab/gemm.o chameleon_recsys_amd/csrc/gemm.hip
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX3_ABL=$v -c -o ab/x3_$v.o chameleon_recsys_amd/csrc/gemm_x3.hip &
done
wait
for v in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ab/libx3_$v.so ab/x3_$v.o ab/gemm.o; done
ls -la ab/*.so
