#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "GSSD_CONV21_WINO=$v"
GSSD_CONV21_WINO=$v python3 scripts/layer_times.py gssdpp f32 2>/dev/null | head -5
done
