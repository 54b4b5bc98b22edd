#!/bin/bash
# round 5: the one-kernel window queries -- parity tests, then the bench step with the debug report
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_minhash_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_wq_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "lockstep_oracle or switches or window_query or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -8 >> gpurun_out/r05_wq_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 2 --warmup 1 $LEAN > gpurun_out/r05_wq_bench.json 2> gpurun_out/r05_wq_bench.log
python3 -m pytest tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -8 >> gpurun_out/r05_wq_tests.log
cat gpurun_out/r05_wq_tests.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_wq_bench.json')); print(j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "one-group slot|part 2 wall" gpurun_out/r05_wq_bench.log | tail -2
