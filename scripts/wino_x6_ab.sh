#!/bin/bash
# Same-box A/B of the fp32-MFMA Winograd kernel and the three-plane one on the trunk shapes (+ parity of both against the direct conv).
# usage (GPU box): bash scripts/wino_x6_ab.sh [extra libs: GSSD_LIB_PATH values to time as well]
cd "$(dirname "$0")/.."
export SHAPES=${SHAPES:-conv3_1:75:32:64,conv3_2:75:64:64,conv4_1:38:64:128,conv4_2:38:128:128,conv5_x:19:128:128}
GSSD_WINO_X6=0 timeout 300 python scripts/bench_wino.py 2>&1 | grep conv | sed "s/direct.*winograd/fp32 winograd/"
GSSD_WINO_X6=1 timeout 300 python scripts/bench_wino.py 2>&1 | grep conv | sed "s/direct.*winograd/x6   winograd/"
for lib in "$@"; do
  echo "== $lib"
  GSSD_LIB_PATH=$PWD/$lib GSSD_WINO_X6=1 timeout 300 python scripts/bench_wino.py 2>&1 | grep conv | sed "s/direct.*winograd/x6   winograd/"
done
