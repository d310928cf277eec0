#!/bin/bash
# round 3, first GPU call: new tests first, then the whole GPU suite, the default bench line, the other configs
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r3a_pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 gpurun_out/r3a_pytest_new.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round3.py > gpurun_out/r3a_pytest_all.log 2>&1; echo "all tests rc=$?"; tail -3 gpurun_out/r3a_pytest_all.log
timeout 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3a_bench.json
timeout 600 python scripts/bench_configs.py > gpurun_out/r3a_configs.jsonl 2> gpurun_out/r3a_configs.err; echo "configs rc=$?"; cat gpurun_out/r3a_configs.jsonl
