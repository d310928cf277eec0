#!/usr/bin/env python3
"""Per-step kernel time of the default bench over time (DVFS / warm-up diagnosis).
usage: python scripts/trace_steps.py [bench args]  -> prints mean kernel ms per block of 50 steps"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TRIRO_BENCH_TRACE="1")
p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-companions", *sys.argv[1:]],
                   capture_output=True, text=True, env=env)
for ln in p.stderr.splitlines():
    if ln.startswith("per-step ms:"):
        xs = [float(x) for x in ln.split()[2:]]
        print("steps", len(xs))
        for i in range(0, len(xs), 50):
            blk = xs[i:i + 50]
            print(f"  {i:5d}-{i + len(blk) - 1:5d}: mean {sum(blk) / len(blk):.4f}  min {min(blk):.4f}  max {max(blk):.4f}")
print(p.stdout.strip()[:400])
