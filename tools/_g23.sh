cd $GRAFT_REPO_ROOT
for v in "" _nolast _nolaned; do
echo "=== variant '$v'"
for i in 1 2; do
MPPI_AMD_LIB=$GRAFT_REPO_ROOT/mppi-generic_amd/lib/libmppi_amd$v.so timeout 600 python -m pytest tests/test_rmppi.py -m gpu -q -k "test_rmppi_rollout_costs_bit_exact and (suspension or complete)" 2>&1 | tail -4
done
done
