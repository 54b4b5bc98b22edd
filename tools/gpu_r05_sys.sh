#!/bin/bash
# round 5: one-launch sketch batches (tests), the systolic DP kernel by range (A/B on the bench step), cfg3 schedules for the auto rule
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_mm_sketch_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_sys_tests.log
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "lockstep_oracle or switches or repeat or oversize or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -6 >> gpurun_out/r05_sys_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for m in 0 3 4 1 0 3; do
NSGPU_KSW_SYS=$m NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_sys_$m.json 2> gpurun_out/r05_sys_$m.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_sys_$m.json')); print('SYS $m', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))" >> gpurun_out/r05_sys_tests.log
grep -E "DP launches by|one-group slot|sketch..chain of" gpurun_out/r05_sys_$m.log | tail -3 >> gpurun_out/r05_sys_tests.log
done
NSGPU_CONS_DEBUG=1 python3 tools/cfg3_sweep.py "256,1,1,4,3" "100,1,1,4,3" "128,1,1,4,3" "100,1,1,3,3" "100,1,2,4,3" "100,1,2,3,2" "160,1,1,4,3" > gpurun_out/r05_cfg3_sweep.txt 2> gpurun_out/r05_cfg3_sweep.log
grep "seed policy" gpurun_out/r05_cfg3_sweep.log >> gpurun_out/r05_cfg3_sweep.txt
cat gpurun_out/r05_sys_tests.log | cut -c1-400; cat gpurun_out/r05_cfg3_sweep.txt
