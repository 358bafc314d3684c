#!/bin/bash
# Knock-out builds of csrc/conv_x6.hip (X6_KO bits: 1 no transform / split, 2 no MFMAs, 4 no weight DMA, 8 no activation loads, 16 no
# fragment reads, 32 no LDS writes) timed on two trunk shapes; results are wrong by construction, only the times mean anything.
# Build here (hipcc cross-compiles): bash scripts/conv_x6_knockout.sh build ; run on the GPU box: bash scripts/conv_x6_knockout.sh run
cd "$(dirname "$0")/.."
CS=grouped-ssd-pytorch_amd/gssd/csrc
KOS="${KOS:-0 1 2 4 8 16 32 3 18 19 27 59}"
if [ "$1" = build ]; then
  mkdir -p build_ko
  for ko in $KOS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -munsafe-fp-atomics -fno-slp-vectorize -DX6_KO=$ko $EXTRA -c $CS/conv_x6.hip -o build_ko/conv_x6_$ko.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/*.o | grep -v '/conv_x6.o$') build_ko/conv_x6_$ko.o -o build_ko/libgssd_ko$ko.so &
  done
  wait
  rm -f build_ko/*.o
else
  for ko in $KOS; do
    echo "== X6_KO=$ko"
    GSSD_LIB_PATH=$PWD/build_ko/libgssd_ko$ko.so python scripts/bench_conv_x6.py ${SHAPES:-conv3_2 conv4_2} 2>&1 | grep -v amdgpu.ids | sed 's/default.*x6:/x6:/'
  done
fi
