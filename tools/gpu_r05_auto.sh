#!/bin/bash
# round 5: the automatic schedule (tests + a third coverage point), the .bsc drop-in test, the quick regression set after the removals
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_ksw2_gpu.py tests/test_mm_sketch_gpu.py tests/test_bwt_gpu.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_auto_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "automatic or lockstep_oracle and not cfg2_full" 2>&1 | tail -12 >> gpurun_out/r05_auto_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "auto:r03" 2>&1 | tail -6 >> gpurun_out/r05_auto_tests.log
NSGPU_CONS_DEBUG=1 python3 tools/mid_sweep.py auto "80,1,3,5,3" "96,1,2,4,3" "96,1,3,5,3" "128,1,2,4,3" "96,1,1,4,3" "128,1,2,5,3" > gpurun_out/r05_mid_sweep.txt 2> gpurun_out/r05_mid_sweep.log
grep -E "seed policy|automatic schedule" gpurun_out/r05_mid_sweep.log >> gpurun_out/r05_mid_sweep.txt
python3 tools/cfg3_sweep.py "128,1,1,4,3" > gpurun_out/r05_cfg3_auto.txt 2>&1
cat gpurun_out/r05_auto_tests.log; cat gpurun_out/r05_mid_sweep.txt gpurun_out/r05_cfg3_auto.txt
