#!/bin/bash
# round 6: one lean step per (graph placement, host threads): what a rank of a shared node gets
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for t in ${THREADS:-2 4 16}; do
  for gph in device host; do
    NSGPU_GRAPH=$gph NSGPU_THREADS=$t NSGPU_CONS_DEBUG=1 timeout 300 python3 bench.py --steps 1 --warmup 0 $LEAN > gpurun_out/r06_thr_${gph}_$t.json 2> gpurun_out/r06_thr_${gph}_$t.log
    python3 -c "import json; j=json.load(open('gpurun_out/r06_thr_${gph}_$t.json')); print('$gph', $t, j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
  done
done
