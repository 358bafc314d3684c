#!/bin/bash
# Knock-out builds of csrc/conv_wino_x6.hip (WX6_KO bits: 1 no vector work, 2 no MFMAs, 4 no U DMA, 8 no patch loads, 16 no fragment reads) timed on
# the trunk shapes; results are wrong by construction, only the times mean anything.
# Build here (hipcc cross-compiles): bash scripts/wino_x6_knockout.sh build ; run on the GPU box: bash scripts/wino_x6_knockout.sh run
cd "$(dirname "$0")/.."
CS=grouped-ssd-pytorch_amd/gssd/csrc
KOS="${KOS:-0 1 2 4 8 16 3 9 11 27 31}"
if [ "$1" = build ]; then
  mkdir -p build_ko
  for ko in $KOS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -munsafe-fp-atomics -fno-slp-vectorize -DWX6_KO=$ko $EXTRA -c $CS/conv_wino_x6.hip -o build_ko/conv_wino_x6_$ko.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/*.o | grep -v '/conv_wino_x6.o$') build_ko/conv_wino_x6_$ko.o -o build_ko/libgssd_wko$ko.so &
  done
  wait
  rm -f build_ko/*.o
else
  for ko in $KOS; do
    echo "== WX6_KO=$ko"
    GSSD_WINO_X6=1 GSSD_LIB_PATH=$PWD/build_ko/libgssd_wko$ko.so ONLY_WINO=1 python scripts/bench_wino.py 2>&1 | grep -v amdgpu.ids | grep "conv" | sed 's/direct.*winograd/winograd/; s/rel err.*//'
  done
fi
