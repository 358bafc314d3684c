#!/bin/bash
# Where does dcn_col2im's time go?  Knock-out builds of csrc/dcn.hip (results wrong on purpose) through GSSD_LIB_PATH.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^dcn.o$')
for ko in 0 1 8 16 24 32 56; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -DCOL2IM_KO=$ko -c dcn.hip -o /tmp/dcn_ko$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dcn_ko$ko.o -o /tmp/libgssd_ko$ko.so
  echo "== COL2IM_KO=$ko   (1 no d(x) flush, 4 no chunk loop, 8 no scatter wave, 16 no reduce waves, 32 no d(cols) DMA)"
  GSSD_LIB_PATH=/tmp/libgssd_ko$ko.so python3 $R/scripts/bench_col2im.py 2>/dev/null | head -3
done
