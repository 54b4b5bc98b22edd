#!/bin/bash
# round 6: lean bench steps with the debug report, consensus graphs in HBM (NSGPU_GRAPH=device), optionally against the pointer graph on the host
# usage: tools/gpu_r06_quick.sh [tag] [ENV=VALUE ...]   (the environment applies to the device runs; tag q: host runs interleaved)
set -x
TAG=${1:-q}; shift
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for i in $(seq 1 ${REPS:-2}); do
  env NSGPU_GRAPH=device "$@" NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r06_${TAG}_dev_$i.json 2> gpurun_out/r06_${TAG}_dev_$i.log
  [ "$TAG" = q ] && NSGPU_GRAPH=host NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r06_${TAG}_host_$i.json 2> gpurun_out/r06_${TAG}_host_$i.log
done
for f in gpurun_out/r06_${TAG}_*.json; do python3 -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"; done
grep -h "graph kernels\|consensus graphs in HBM" gpurun_out/r06_${TAG}_dev_1.log | tail -8
