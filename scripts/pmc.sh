#!/bin/bash
# usage: scripts/pmc.sh <outdir> <counter list> -- <python script args...>   (run on the GPU box)
out=$1; shift; ctr=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -o p -- python3 "$@" > /dev/null 2>&1
python3 - "$GRAFT_REPO_ROOT/gpurun_out/$out" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'conv_' not in k: continue
    print(k, {c: round(sum(v[len(v)//2:]) / len(v[len(v)//2:])) for c, v in d.items()}, 'n=', len(next(iter(d.values()))))
PY
