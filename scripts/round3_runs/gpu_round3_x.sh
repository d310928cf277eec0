#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python scripts/bench_build.py 2>/dev/null | cut -c1-260
timeout 200 python scripts/bench_build.py --opt node_layout=0 2>/dev/null | cut -c1-260
