#!/bin/bash
# Copies what scripts/gpu_final_r06.sh left in gpurun_out/r06/ (merged back from the GPU box) into profiles/ under the round's names.
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/${TAG:-r06}
for f in bench.json bench_profiled.json kernel_stats.csv kernel_trace_step.txt pmc_traffic.json h2_sq_counters.txt h2_blocked_microbench.txt gemm_breakdown.txt \
         host_profile.txt emulated_rank_of_n.txt shard_kernel_stats.csv shard_kernel_trace_step.txt stress_kernel_stats.csv stress_profiled.json \
         bf16_kernel_stats.csv bf16_profiled.json ab_arms.txt smoke.log pytest_gpu.log loss_curve_200_A.json loss_curve_200_B.json loss_curve_200_C.json \
         loss_curve_bf16_50.json; do
  [ -s $S/$f ] && cp $S/$f profiles/r06_$f && echo "profiles/r06_$f"
done
