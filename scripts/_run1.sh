for i in 1 2 3; do timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "rel err"; done
GSSD_NO_GEMM_SLOT=1 timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "rel err"
