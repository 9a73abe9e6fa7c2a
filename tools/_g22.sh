cd $GRAFT_REPO_ROOT
for v in s1c3 s1c3aw s1c3 s1c3aw; do MPPI_AMD_LIB=$GRAFT_REPO_ROOT/mppi-generic_amd/lib/libmppi_amd_$v.so python tools/_rl_ar.py; done
