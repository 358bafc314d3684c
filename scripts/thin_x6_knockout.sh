#!/bin/bash
# knock-out timing of csrc/conv_thin_x6.hip (make -C grouped-ssd-pytorch_amd/gssd/csrc thin_x6_ko): which part of the tile loop costs what
cd "$(dirname "$0")/.."
echo "== full kernel"; python scripts/bench_thin_x6.py | grep conv
for k in 1 2 4 6 8 16 17 31; do
  f=grouped-ssd-pytorch_amd/gssd/lib/libgssd_hip_tx6ko$k.so
  [ -f $f ] || continue
  echo "== TX6_KO=$k (1 no transform/split, 2 no MFMAs, 4 no fragment reads, 8 no stores, 16 no loads)"
  GSSD_LIB_PATH=$PWD/$f python scripts/bench_thin_x6.py | grep conv
done
