#!/bin/bash
# round 6, GPU job 12: (1) the native step after the peers' sends moved to a side stream + the watchdog: loopback test, and a
# receive that nobody answers ended by tr_comm_abort (each in a child process under a timeout); (2) A/B: the drain of parked
# leaf tests marked as a COLD branch (the registers the float64 call clobbers are saved around the call instead of for the
# whole kernel: no scratch traffic in the prologue) against the shipped library
mkdir -p gpurun_out
timeout -s KILL 300 python tests/native_step_world1.py > gpurun_out/r06_native12.txt 2>&1; echo "native_step rc=$?" >> gpurun_out/r06_native12.txt
timeout -s KILL 150 python tests/native_abort_world1.py > gpurun_out/r06_abort12.txt 2>&1; echo "native_abort rc=$?" >> gpurun_out/r06_abort12.txt
tail -3 gpurun_out/r06_native12.txt; grep -v "^$" gpurun_out/r06_abort12.txt | tail -6
python -c "import torch; x=torch.ones(4,device='cuda'); print('gpu alive', float(x.sum()))"
AB_SET=direct timeout 900 bash scripts/round5/ab.sh gpurun_out/r06_ab12.txt base cold base cold > gpurun_out/r06_ab12.log 2>&1
AB_SET=stream timeout 600 bash scripts/round5/ab.sh gpurun_out/r06_ab12s.txt base cold > gpurun_out/r06_ab12s.log 2>&1
cat gpurun_out/r06_ab12.txt gpurun_out/r06_ab12s.txt
