#!/bin/bash
# round 5: with class 12 on <4,1>, where should the long problems of the one-wave classes start?  NSGPU_KSW_LONG_ROWS sweep, interleaved
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -3
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for i in 1 2; do
for v in 900 650 450 300; do
NSGPU_KSW_LONG_ROWS=$v NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_longrows_${v}_$i.json 2> gpurun_out/r05_longrows_${v}_$i.log
python3 -c "import json; j=json.load(open('gpurun_out/r05_longrows_${v}_$i.json')); print('long rows $v:', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
grep -E "DP launches by|one-group slot" gpurun_out/r05_longrows_${v}_$i.log | tail -2
done
done
