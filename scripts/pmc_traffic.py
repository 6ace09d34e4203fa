#!/usr/bin/env python
"""Per-kernel-symbol HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; CSV output):
traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes).  The x2 is MI355X_MICROARCH.md's gfx950 correction for wide coalesced
reads (FETCH_SIZE tallies 128-B requests at 64 B); WRITE_SIZE matched the algorithmic output bytes exactly on this path.
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/rNN_pmc_traffic.json"""
import collections
import csv
import json
import os
import sys


def per_symbol(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
    return {k: tot[k] / n[k] for k in tot}, n


f, nf = per_symbol(sys.argv[1], 'FETCH_SIZE')
w, _ = per_symbol(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in f:
    out[k] = dict(dispatches=nf[k], fetch_kb_raw=round(f[k], 1), write_kb=round(w.get(k, 0.0), 1),
                  traffic_bytes_per_launch=round((2 * f[k] + w.get(k, 0.0)) * 1024))
out["_generated_by"] = "%s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 2 --warmup 1 --no-arms ...` -> scripts/pmc_traffic.py" % os.environ.get("PMC_GENERATED_BY", "scripts/pmc_traffic.py")
json.dump(out, sys.stdout, indent=1, sort_keys=True)
