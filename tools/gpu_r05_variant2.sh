#!/bin/bash
# round 5: wave shapes of the BULK DP classes (NSGPU_KSW_VARIANT bits: 1 = class 1 on <4,1>, 2 = class 0 on <2,1>, 4 = class 1 on <2,2>): parity, then A/B
set -x
mkdir -p gpurun_out
NSGPU_KSW_VARIANT=3 python3 -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -3
NSGPU_KSW_VARIANT=4 python3 -m pytest tests/test_ksw2_gpu.py -m gpu -x -q 2>&1 | tail -3
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for i in 1 2; do
for v in 0 1 4 2 3; do
NSGPU_KSW_VARIANT=$v NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_variant2_${v}_$i.json 2> gpurun_out/r05_variant2_${v}_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_variant2_${v}_$i.json')); print('cfg2 variant $v:', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "DP launches by|one-group slot" gpurun_out/r05_variant2_${v}_$i.log | tail -2
done
done
