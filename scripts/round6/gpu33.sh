#!/bin/bash
# round 6, GPU job 33: the fuzz with the new family of exact ties (integer-grid height fields under lattice rays through
# their vertices, edges and diagonals): 300 iterations of that family alone, 200 of the mix; and the fixed-seed fuzz test of the suite
mkdir -p gpurun_out
timeout 1500 python scripts/fuzz_parity.py --iters 300 --seed 661 --kind 5 > gpurun_out/r06_fuzz_grid661.txt 2>&1; tail -2 gpurun_out/r06_fuzz_grid661.txt | cut -c1-300
timeout 1200 python scripts/fuzz_parity.py --iters 200 --seed 662 > gpurun_out/r06_fuzz_seed662.txt 2>&1; tail -1 gpurun_out/r06_fuzz_seed662.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k fuzz 2>&1 | tail -2
