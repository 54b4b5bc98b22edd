#!/bin/bash
# round 5: the widest DP class from device lists + deferred alignments + parallel host index builds: regression set, repeats genome, cfg2
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py tests/test_seeds_gpu.py tests/test_chain_gpu.py tests/test_plan_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_cls3_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "deferred or repeat or lockstep_oracle or switches or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -8 >> gpurun_out/r05_cls3_tests.log
cat gpurun_out/r05_cls3_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for i in 1 2; do
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 0 $LEAN --genome repeats > gpurun_out/r05_cls3_rep_$i.json 2> gpurun_out/r05_cls3_rep_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_cls3_rep_$i.json')); print(j['value'], j['ms_per_step'], j['compression'])"
grep -E "deferred alignments|pairs handed back|one-group slot|sketch..chain of|left to the host's plan|DP launches by|part 2 wall" gpurun_out/r05_cls3_rep_$i.log
done
for i in 1 2; do
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_cls3_cfg2_$i.json 2> gpurun_out/r05_cls3_cfg2_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_cls3_cfg2_$i.json')); print(j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "one-group slot" gpurun_out/r05_cls3_cfg2_$i.log | tail -1
done
