"""Critical-path view of one training step from a rocprofv3 kernel trace (CSV): per-queue timeline of the LAST full step,
the intervals in which no MFMA GEMM is running, and what runs in them.  usage: timeline.py <kernel_trace.csv> [step_marker]"""
import csv
import os
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_sel_count_valid"      # first sampler kernel of a step
starts = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
# the sampler may launch the marker more than once per step: keep markers separated by > 5 ms
steps = []
for i in starts:
    if not steps or rows[i]['s'] - rows[steps[-1]]['s'] > float(os.environ.get('TIMELINE_MIN_GAP_NS', 5e6)):
        steps.append(i)
a, b = steps[-3], steps[-2]
step = rows[a:b]
t0 = step[0]['s']
print("step wall %.3f ms, %d kernels, sum of kernel time %.3f ms" % ((rows[b]['s'] - t0) / 1e6, len(step), sum(r['e'] - r['s'] for r in step) / 1e6))
isg = lambda r: any(k in r['Kernel_Name'] for k in ('gemm_f32_kernel', 'gemm_bf16_kernel', 'gemm_x3_kernel', 'gemm_b16_kernel', 'gemm_p3_kernel', 'gemm_h2_kernel', 'gemm_h2w_kernel', 'gemm_b1_kernel'))
# union of GEMM-active intervals, clipped to the step's window (a kernel of the previous step's tail may still run at its head, one of
# this step's tail beyond its end: the next step's sampler starts under the W2 weight gradient)
tend = rows[b]['s']
iv = sorted((max(r['s'], t0), min(r['e'], tend)) for r in rows if isg(r) and r['e'] > t0 and r['s'] < tend)
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged)
print("GEMM-active %.3f ms; GEMM-idle %.3f ms" % (busy / 1e6, (rows[b]['s'] - t0 - busy) / 1e6))
gaps = []
prev = t0
for s, e in merged + [[rows[b]['s'], rows[b]['s']]]:
    if s - prev > 20e3:
        gaps.append((prev, s))
    prev = max(prev, e)
print("GEMM-idle gaps > 20 us:")
for s, e in gaps:
    inside = [r for r in step if r['e'] > s and r['s'] < e and not isg(r)]
    names = {}
    for r in inside:
        n = r['Kernel_Name'].split('(')[0][-48:]
        names[n] = names.get(n, 0) + (min(r['e'], e) - max(r['s'], s)) / 1e3
    top = sorted(names.items(), key=lambda kv: -kv[1])[:6]
    print("  +%.3f ms  len %.3f ms: %s" % ((s - t0) / 1e6, (e - s) / 1e6, ", ".join("%s %.0fus" % kv for kv in top)))
if len(sys.argv) > 3:
    for r in step:
        print("%8.3f %8.3f q%s %s" % ((r['s'] - t0) / 1e6, (r['e'] - r['s']) / 1e6, r['Queue_Id'], r['Kernel_Name'][:90]))
