#!/usr/bin/env python3
"""one-line digest of a bench.py JSON line read from stdin"""
import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
print(f"{d['value'] / 1e9:8.3f} G/s  {d['ms_per_step']:8.3f} ms  frac={r['frac']:.4f} "
      f"avg_launch={r['avg_launch_ms']:.4f} ms share={r['kernel_share_of_step']:.3f}")
