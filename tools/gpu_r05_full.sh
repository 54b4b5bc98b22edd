#!/bin/bash
# round 5: the whole GPU suite as the driver runs it, then BASELINE configs[3]'s per-GPU share on this one GPU (memory first: what the box has)
set -x
mkdir -p gpurun_out
(free -g; cat /sys/fs/cgroup/memory.max 2>/dev/null; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > gpurun_out/r05_box.txt 2>&1
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_full_tests.log
cat gpurun_out/r05_full_tests.log
timeout 900 python3 tools/cfg4_share.py 1 200000 10000 > gpurun_out/r05_cfg4_200k.txt 2>&1
cat gpurun_out/r05_box.txt gpurun_out/r05_cfg4_200k.txt | tail -20
