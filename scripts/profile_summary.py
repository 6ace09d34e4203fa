#!/usr/bin/env python
"""Builds profiles/<tag>_summary.md from the committed artifacts of one build: bench JSON lines, rocprofv3 kernel stats (CSV),
PMC traffic JSON.  usage: profile_summary.py r01j > profiles/r01j_summary.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1]
    b = json.load(open(os.path.join(P, tag + "_bench.json")))
    bp = json.load(open(os.path.join(P, tag + "_bench_profiled.json")))
    r = b["roofline"]
    out = ["# %s - summary of the committed measurement artifacts (1 x MI355X)\n" % tag]
    out.append("| | |\n|---|---|")
    out.append("| command | `python bench.py --gpus 1 --steps %d --warmup %d` (`%s_bench.json`) |" % (b["steps"], b["warmup"], tag))
    out.append("| %s | **%.0f %s**, %.3f ms/step, dtype %s, %s |" % (b["metric"], b["value"], b["unit"], b["ms_per_step"], b["dtype"], b["config"]["workload"]))
    g = b.get("g1_like_session_lengths")
    if g:
        out.append("| G1-like session lengths | %.0f sessions/s, %.3f ms/step, %.1f %% of the padded positions valid |" % (g["value"], g["ms_per_step"], 100 * g["valid_positions_of_padded"]))
    out.append("| roofline (%s) | %s: achieved %.1f %s of %.1f = **%.3f**; %d launch(es)/step, %.3f ms each (HIP events), %.3f GFLOP algorithmic |" % (
        r["bound"], r["kernel"].split("(GemmParams)")[0].replace("void ", ""), r["achieved"], r["unit"], r["peak"], r["frac"], r["launches_per_step"], r["avg_launch_ms"], r["algorithmic_gflop_per_launch"]))
    out.append("| same kernel under rocprofv3 | HIP events %.3f ms in the profiled run (`%s_bench_profiled.json`); `--stats` average below |" % (bp["roofline"]["avg_launch_ms"], tag))
    if r.get("traffic"):
        out.append("| HBM traffic (PMC) | %.2f GB per launch vs %.2f GB algorithmic (`%s`) |" % (r["traffic"] / 1e9, r["algorithmic_bytes_per_launch"] / 1e9, r["traffic_source"]))
    c = b.get("cpu_baseline")
    if c:
        out.append("| CPU baseline (%s) | %.1f %s on %d threads - %s |" % (c["kind"], c["value"], c["unit"], c["cores"], c["sample"]))
    a = b.get("accuracy_vs_cpu_ref")
    if a:
        out.append("| HitRate@5 / MRR@5 | %s / %s |" % (json.dumps(a["hitrate_at_5"]), json.dumps(a["mrr_at_5"])))
    for name, label in (("bf16", "bf16 compute mode"), ("adressa", "Adressa shape"), ("g1_padded_masked", "G1-like lengths, padded + masked (CHAM_COMPACT=0)")):
        fn = os.path.join(P, "%s_%s_bench.json" % (tag, name))
        if os.path.exists(fn):
            x = json.load(open(fn))
            out.append("| %s | %.0f sessions/s, %.3f ms/step (`%s`) |" % (label, x["value"], x["ms_per_step"], os.path.basename(fn)))
    out.append("\n## rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg` (`%s_kernel_stats.csv`)\n" % tag)
    rows = list(csv.DictReader(open(os.path.join(P, tag + "_kernel_stats.csv"))))
    # optimizer steps in the profiled run = launches of the Adam kernel (warm-up + timed + host-enqueue + roofline-leg steps)
    steps = float(next((int(x["Calls"]) for x in rows if x["Name"].startswith("k_adam_tf")), 16))
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    out.append("Sum of kernel time %.2f ms/step over %d kernel symbols (lanes overlap: wall time per step is lower).\n" % (tot / 1e6 / steps, len(rows)))
    out.append("| kernel | launches/step | ms/step | avg us | share |\n|---|---|---|---|---|")
    traffic = {}
    tf = os.path.join(P, tag + "_pmc_traffic.json")
    if os.path.exists(tf):
        traffic = {k: v for k, v in json.load(open(tf)).items() if isinstance(v, dict)}
    for x in rows[:16]:
        out.append("| `%s` | %.1f | %.3f | %.1f | %.1f %% |" % (x["Name"].split("(")[0].replace("void ", "")[:80], int(x["Calls"]) / steps,
                                                             float(x["TotalDurationNs"]) / 1e6 / steps, float(x["AverageNs"]) / 1e3, float(x["Percentage"])))
    if traffic:
        out.append("\n## HBM traffic per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate passes; `%s_pmc_traffic.json`)\n" % tag)
        out.append("| kernel | GB per launch | fetch (raw KB) | write (KB) |\n|---|---|---|---|")
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:10]:
            out.append("| `%s` | %.3f | %.0f | %.0f |" % (k.split("(")[0].replace("void ", "")[:80], v["traffic_bytes_per_launch"] / 1e9, v["fetch_kb_raw"], v["write_kb"]))
    print("\n".join(out))


if __name__ == "__main__":
    main()
