#!/bin/bash
# round 6: the bench's host-threads sweep (children of a bench process that holds the GPU, graphs in HBM by the automatic placement) again and again
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --graph-leg 0 --cfg3-leg 0"
for i in $(seq 1 ${1:-5}); do
  timeout 600 python3 bench.py --steps 1 --warmup 0 $LEAN --threads-sweep 1 > gpurun_out/r06_sws_$i.json 2> gpurun_out/r06_sws_$i.log
  python3 -c "import json; j=json.load(open('gpurun_out/r06_sws_$i.json')); print($i, j['value'], [(r.get('host_threads'), r.get('value'), r.get('lossless_roundtrip_bad_reads'), r.get('streams_identical_to_the_fixture'), r.get('error')) for r in j['host_threads_sweep']['runs']])"
done
