#!/bin/bash
# Local pre-flight for a GPU-box call: rebuild the in-tree libraries when a source changed (the .so travels with the snapshot - a
# stale one would be measured), run the CPU test suite, then hand the command to gpurun.
#   bash scripts/run_gpu.sh [--timeout S] '<command run on the GPU box>'
set -e
cd "$(dirname "$0")/.."
TO=1500
if [ "$1" = "--timeout" ]; then TO=$2; shift 2; fi
python -c 'import __graft_entry__ as g; g.build()'
timeout 600 python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -2
/usr/local/graft/bin/gpurun --timeout $TO -- "$1"
