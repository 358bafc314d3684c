#!/bin/bash
# Same-box A/B of flash_attn_x6 with the next tile's DMA pieces at the tile's top (-DFX6_SPREAD=0) and dealt out between the MFMA groups (default).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^flash_attn_x6.o$')
for v in 0 1 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize -DFX6_SPREAD=$v -c flash_attn_x6.hip -o /tmp/fx6_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/fx6_$v.o -o /tmp/libgssd_fx6_$v.so
  echo "== FX6_SPREAD=$v"
  GSSD_LIB_PATH=/tmp/libgssd_fx6_$v.so python3 $R/scripts/bench_flash_x6.py 2>/dev/null | tail -1
done
