#!/bin/bash
# round 5: wave shapes of the late DP classes (NSGPU_KSW_VARIANT bits: 1 = class 12 on <4,1>, 2 = class 8 on <12,1>, 4 = class 3 on <16,3>): parity, then A/B
set -x
mkdir -p gpurun_out
NSGPU_KSW_VARIANT=7 python3 -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_variant_tests.log
cat gpurun_out/r05_variant_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for i in 1 2; do
for v in 0 1 2 3; do
NSGPU_KSW_VARIANT=$v NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_variant_cfg2_${v}_$i.json 2> gpurun_out/r05_variant_cfg2_${v}_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_variant_cfg2_${v}_$i.json')); print('cfg2 variant $v:', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "DP launches by" gpurun_out/r05_variant_cfg2_${v}_$i.log | tail -1
done
done
for v in 0 4 7; do
NSGPU_KSW_VARIANT=$v NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 0 $LEAN --genome repeats > gpurun_out/r05_variant_rep_$v.json 2> gpurun_out/r05_variant_rep_$v.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_variant_rep_$v.json')); print('repeats variant $v:', j['value'], j['ms_per_step'])"
grep -E "DP launches by" gpurun_out/r05_variant_rep_$v.log | tail -1
done
