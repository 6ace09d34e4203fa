#!/bin/bash
# the tile-blocked plane layout with different paddings behind each 16 KB block (H2B_PAD elements): stand-alone GEMMs + producers per build
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${TAG:-r06pad}; mkdir -p $O; cd $R; export PYTHONPATH=$R
for pad in ${PADS:-0 128 512 2048}; do
  CHAM_BUILD_DEFINES="-DH2B_PAD=$pad" python -m chameleon_recsys_amd.build > $O/build_$pad.log 2>&1
  CHAM_BUILD_DEFINES="-DH2B_PAD=$pad" timeout 300 python scripts/bench_h2_blocked.py > $O/pad_$pad.txt 2>&1
  cat $O/pad_$pad.txt
done
