cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_rmppi.py tests/test_double_integrator_robust_cost.py tests/test_long_horizon.py -m gpu -q 2>&1 | tail -8
python tools/_rl_ar.py
python tools/robust_latency.py
