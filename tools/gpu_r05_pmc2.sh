#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests/test_consensus_gpu.py tests/test_dist_gpu.py -m gpu -x -q -k "lockstep_oracle and not cfg2_full or switches or cxx_driver or two_ranks" 2>&1 | tail -5 > gpurun_out/r05_split_tests.log
cat gpurun_out/r05_split_tests.log
bash tools/gpu_r05_pmc.sh
cd "$GRAFT_REPO_ROOT"
NSGPU_CONS_DEBUG=1 python3 bench.py --genome repeats --steps 1 --warmup 0 --throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 > gpurun_out/r05_repeats.json 2> gpurun_out/r05_repeats.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_repeats.json')); print('repeats', j['value'], j['ms_per_step'])"
grep -E "one-group slot|window-query batches|sketch..chain of" gpurun_out/r05_repeats.log | tail -3
