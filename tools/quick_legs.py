import sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import bench
for name, fn in (("autorally", lambda: bench.autorally_leg(0, with_cpu_baseline=False)), ("lstm_colored", lambda: bench.lstm_colored_leg(0))):
    r = fn()
    print(name, r["ms_per_step"], r["roofline"]["avg_kernel_us"], r["roofline"]["frac"])
