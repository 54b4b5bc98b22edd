#!/bin/bash
# round 5: the seeding rewrite (persistent count tables + streaming seed kernel): its tests, the engine under NSGPU_SKETCH_CHECK, a lean bench
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_seeds_gpu.py tests/test_align_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_seeds_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "lockstep_oracle or switches or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -15 >> gpurun_out/r05_seeds_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 2 --warmup 1 $LEAN > gpurun_out/r05_seeds_bench.json 2> gpurun_out/r05_seeds_bench.log
cat gpurun_out/r05_seeds_tests.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_seeds_bench.json')); print(j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "one-group slot|part 2 wall|sketch\.\.chain|seeds" gpurun_out/r05_seeds_bench.log | tail -8
