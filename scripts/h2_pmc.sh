#!/bin/bash
# SQ / TCC counter passes over the two-fp16-plane GEMM micro-benchmark (tests/bench_gemm_h2.py, H2_ONLY=1: the default kernels only).
# gpurun from the repo root; separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $R && H2_ONLY=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/h2pmc_$i -o p$i -- python -m tests.bench_gemm_h2 >/dev/null 2>$O/h2pmc_$i.err )
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(list)
for f in glob.glob("$O/h2pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'gemm_h2' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']);
        cnt[(k,r['Counter_Name'])]+=1
for f in glob.glob("$O/h2pmc_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'gemm_h2' in k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,d in sorted(agg.items()):
    print(k, "  kernel-trace durations (ms, under the counter pass):", ["%.3f"%x for x in dur.get(k,[])][:8])
    for c,v in sorted(d.items()): print("   %-34s %16.0f  (per launch %14.0f, %d launches)"%(c,v,v/max(1,cnt[(k,c)]),cnt[(k,c)]))
PY
rm -rf $O/h2pmc_*/
