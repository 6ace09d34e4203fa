#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.log
( timeout 300 python -m tests.bench_gemm 0 1 2 3 4 2>&1 | tail -30 ) > $O/gemm_variants.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 ) > $O/bench.log
( CHAM_OVERLAP=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_overlap.log
( CHAM_OVERLAP=1 CHAM_RNN_LDS_HOG=120000 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_overlap_hog.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o r01b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 ) > $O/rocprof.log
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmc_sq.log
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmc_write.log
cd $R
find $O -name '*kernel_trace*' -size +30M -delete
ls -R $O | head -50
cat $O/pytest_gpu.log $O/gemm_variants.log $O/bench.log $O/bench_overlap.log $O/bench_overlap_hog.log
