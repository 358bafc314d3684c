#!/bin/bash
# Same-box A/B of the BatchNorm-backward streaming kernels' unroll factor / occupancy target (csrc/backward.hip: BN_UNR, BN_OCC).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^backward.o$')
for v in "1 4" "1 8" "2 8" "4 4"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -DBN_UNR=$1 -DBN_OCC=$2 -c backward.hip -o /tmp/bw_$1_$2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/bw_$1_$2.o -o /tmp/libgssd_bw_$1_$2.so
  for dt in f32 bf16; do
    GSSD_LIB_PATH=/tmp/libgssd_bw_$1_$2.so python3 $R/bench.py --dtype $dt --steps 10 --warmup 3 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --full-step 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('BN_UNR=$1 BN_OCC=$2 $dt full step', d['full_step']['ms_per_step'])"
  done
done
