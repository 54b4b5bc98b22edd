#!/bin/bash
# round 5: the second early exit -- parity tests, then the bench step A/B
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_ksw2_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_mte_tests.log
python3 -m pytest tests/test_align_gpu.py tests/test_consensus_gpu.py -m gpu -x -q -k "align or lockstep_oracle and not cfg2_full or one_builder_equals_oracle and not cfg2_full" 2>&1 | tail -8 >> gpurun_out/r05_mte_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for v in on off on off; do
  if [ $v = off ]; then export NSGPU_KSW_KEEP_MTE=1; else unset NSGPU_KSW_KEEP_MTE; fi
  NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_mte_$v.json 2> gpurun_out/r05_mte_$v.log
  python3 -c "import json; j=json.load(open('gpurun_out/r05_mte_$v.json')); print('MTE exit $v', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))" >> gpurun_out/r05_mte_tests.log
  grep -E "DP launches by|one-group slot" gpurun_out/r05_mte_$v.log | tail -2 >> gpurun_out/r05_mte_tests.log
done
unset NSGPU_KSW_KEEP_MTE
python3 tools/bench_ksw_rows.py 0x100040 > gpurun_out/r05_rows_ext_mte.txt 2>&1
cat gpurun_out/r05_mte_tests.log | cut -c1-330
