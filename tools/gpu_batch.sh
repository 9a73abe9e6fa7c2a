set -x
python tools/ab_kernels.py robust_racer_all robust_racer racer 2>&1 | tail -n 8
timeout 900 python -m pytest tests/test_rmppi.py tests/test_full_size_parity.py tests/test_handover.py tests/test_racer_dubins_suspension.py tests/test_racer_dubins_lstm_unc.py -m gpu -x -q 2>&1 | tail -n 4
