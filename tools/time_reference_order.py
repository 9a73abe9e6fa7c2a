#!/usr/bin/env python3
"""iteration time with the reference-order reduction (bench.py: reference_order_leg) — quick A/B aid"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import bench
print(json.dumps(bench.reference_order_leg(0)))
