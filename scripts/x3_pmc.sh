#!/bin/bash
# SQ counter passes over the bf16x3 GEMM micro-benchmark (X3_ONLY=1: only the bf16x3 variants of the CAR dgrad shape).  gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  ( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/x3pmc_$i -o p$i -- python -m tests.bench_gemm_x3 >/dev/null 2>$O/x3pmc_$i.err )
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$O/x3pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        if 'gemm' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        cnt[(k,r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-34s %16.0f  (per launch %14.0f)"%(c,v,v/max(1,cnt[(k,c)])))
PY
