cd $GRAFT_REPO_ROOT
for v in _allw; do
echo "=== variant '$v'"
MPPI_AMD_LIB=$GRAFT_REPO_ROOT/mppi-generic_amd/lib/libmppi_amd$v.so timeout 600 python -m pytest tests/test_rmppi.py -m gpu -q -k "test_rmppi_rollout_costs_bit_exact and (suspension)" 2>&1 | tail -4
MPPI_AMD_LIB=$GRAFT_REPO_ROOT/mppi-generic_amd/lib/libmppi_amd$v.so DBG_T=2 python tools/_dbg_rm.py suspension 0 injected 2>&1 | grep -v "first idx\|^\["
done
