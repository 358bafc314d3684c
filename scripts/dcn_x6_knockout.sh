#!/bin/bash
# Where does dcn_x6's time go?  Knock-out builds (results wrong on purpose) through GSSD_LIB_PATH.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^dcn_x6.o$')
for ko in ${KOS:-0 1 2 4 8 9 3 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize $EXTRA -DX6_KO=$ko -c dcn_x6.hip -o /tmp/x6_ko$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/x6_ko$ko.o -o /tmp/libgssd_x6_$ko.so
  echo "== $EXTRA X6_KO=$ko  (1 no blend / split VALU, 2 no MFMAs, 4 no weight DMA, 8 no x loads, 16 no fragment reads, 32 no barrier)"
  GSSD_LIB_PATH=/tmp/libgssd_x6_$ko.so python3 $R/scripts/bench_dcn_x6.py 2>/dev/null | tail -1
done
