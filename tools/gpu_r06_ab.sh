#!/bin/bash
# round 6: the same lean device-graph step under several settings, one after the other on one box
# usage: tools/gpu_r06_ab.sh "ENV=V ENV2=V" "ENV=W" ...   (each argument one setting; "-" = defaults)
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
i=0
for rep in 1 2; do
for setting in "$@"; do
  i=$((i+1))
  s="$setting"; [ "$s" = "-" ] && s=""
  env NSGPU_GRAPH=device NSGPU_CONS_DEBUG=1 $s python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r06_ab_$i.json 2> gpurun_out/r06_ab_$i.log
  python3 -c "import json; j=json.load(open('gpurun_out/r06_ab_$i.json')); print('[$setting]', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
  grep -h "consensus graphs in HBM\|by duration\|ms in sum by phase" gpurun_out/r06_ab_$i.log | tail -3 | cut -c1-420
done
done
