#!/bin/bash
# round 5: idle threads of the early-update watch take over other threads' builders (NSGPU_CONS_NO_STEAL=1 = owners only): parity, then A/B
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_consensus_gpu.py tests/test_stress_gpu.py -m gpu -x -q -k "deferred or repeat or lockstep_oracle or switches or stress or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -5
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for i in 1 2 3; do
for v in steal nosteal; do
if [ $v = nosteal ]; then export NSGPU_CONS_NO_STEAL=1; else unset NSGPU_CONS_NO_STEAL; fi
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_steal_${v}_$i.json 2> gpurun_out/r05_steal_${v}_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_steal_${v}_$i.json')); print('$v:', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "graph updates run ahead|one-group slot" gpurun_out/r05_steal_${v}_$i.log | tail -2
done
done
unset NSGPU_CONS_NO_STEAL
for v in steal nosteal; do
if [ $v = nosteal ]; then export NSGPU_CONS_NO_STEAL=1; else unset NSGPU_CONS_NO_STEAL; fi
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 0 $LEAN --genome repeats > gpurun_out/r05_steal_rep_$v.json 2> gpurun_out/r05_steal_rep_$v.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_steal_rep_$v.json')); print('repeats $v:', j['value'], j['ms_per_step'])"
done
