#!/bin/bash
# round 5: the two-lane slot -- regression subset, A/B against one batch per slot, per-row DP numbers
set -x
mkdir -p gpurun_out
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "lockstep_oracle or switches or one_builder_equals_oracle and not cfg2_full_one and not cfg3_at_size" 2>&1 | tail -8 > gpurun_out/r05_lanes_tests.log
python3 -m pytest tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -8 >> gpurun_out/r05_lanes_tests.log
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
for rep in 1 2; do
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_lanes_on_$rep.json 2> gpurun_out/r05_lanes_on_$rep.log
NSGPU_NO_LANES=1 NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_lanes_off_$rep.json 2> gpurun_out/r05_lanes_off_$rep.log
done
python3 tools/bench_ksw_rows.py 0x80008 > gpurun_out/r05_rows_fill.txt 2>&1
python3 tools/bench_ksw_rows.py 0x40 > gpurun_out/r05_rows_ext.txt 2>&1
NSGPU_KSW_SYS=2 python3 tools/bench_ksw_rows.py 0x80008 > gpurun_out/r05_rows_fill_sys.txt 2>&1
cat gpurun_out/r05_lanes_tests.log
for f in gpurun_out/r05_lanes_o*.json; do python3 -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"; done
