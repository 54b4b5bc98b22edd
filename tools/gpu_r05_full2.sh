#!/bin/bash
# round 5: the whole GPU suite as the driver runs it, then one lean step with the debug report
set -x
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_full2_tests.log
cat gpurun_out/r05_full2_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 2 --warmup 1 $LEAN > gpurun_out/r05_full2_bench.json 2> gpurun_out/r05_full2_bench.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_full2_bench.json')); print(j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "graph updates run ahead|one-group slot|early tasks" gpurun_out/r05_full2_bench.log | tail -3
