#!/bin/bash
# round 6: two independent bench processes on ONE GPU at the same time, consensus graphs in HBM in both: every run must round-trip and equal the fixture
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for i in $(seq 1 ${1:-3}); do
  for p in a b; do
    NSGPU_GRAPH=device NSGPU_THREADS=${2:-4} timeout 400 python3 bench.py --steps 1 --warmup 0 $LEAN > gpurun_out/r06_two_${i}$p.json 2> gpurun_out/r06_two_${i}$p.log &
  done
  wait
  for p in a b; do python3 -c "
import json
try:
    j=json.load(open('gpurun_out/r06_two_${i}$p.json')); print('$i$p', j['value'], j['ms_per_step'], 'bad reads', j['config']['lossless_roundtrip_bad_reads'], 'parity', j.get('parity',{}).get('all_identical'))
except Exception as ex:
    print('$i$p', 'no result:', open('gpurun_out/r06_two_${i}$p.log').read()[-400:].replace(chr(10),' | '))
"; done
done
