#!/bin/bash
# round 5: BASELINE configs[3]'s per-GPU share on one GPU (2 Gbases first, then the whole 6.25 Gbases), the 2-rank bench test with the automatic count
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_cfg4_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "switches or window_query" 2>&1 | tail -4 >> gpurun_out/r05_cfg4_tests.log
timeout 900 python3 tools/cfg4_share.py 1 200000 10000 > gpurun_out/r05_cfg4_200k.txt 2>&1
tail -3 gpurun_out/r05_cfg4_200k.txt
if grep -q '"bad_reads": 0' gpurun_out/r05_cfg4_200k.txt; then
  NSGPU_CONS_DEBUG=1 timeout 1500 python3 tools/cfg4_share.py 2 625000 10000 > gpurun_out/r05_cfg4_625k.txt 2> gpurun_out/r05_cfg4_625k.log
fi
cat gpurun_out/r05_cfg4_tests.log; tail -4 gpurun_out/r05_cfg4_625k.txt; grep -E "automatic schedule|resident|process CPU" gpurun_out/r05_cfg4_625k.log | tail -4
