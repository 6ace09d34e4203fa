#!/bin/bash
# Copies what scripts/gpu_final_r05.sh left in gpurun_out/r05/ (merged back from the GPU box) into profiles/ under the round's names.
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/${TAG:-r05}
for f in bench.json bench_profiled.json kernel_stats.csv kernel_trace_step.txt pmc_traffic.json h2_sq_counters.txt gemm_h2_microbench.txt host_profile.txt \
         emulated_rank_of_n.txt shard_kernel_stats.csv shard_kernel_trace_step.txt stress_kernel_stats.csv stress_profiled.json smoke.log pytest_gpu.log; do
  [ -s $S/$f ] && cp $S/$f profiles/r05_$f && echo "profiles/r05_$f"
done
