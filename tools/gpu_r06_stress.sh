#!/bin/bash
# round 6: the same device-graph step again and again with a given number of host threads: every run must round-trip and equal the fixture
# usage: tools/gpu_r06_stress.sh THREADS RUNS [ENV=V ...]
T=$1; N=$2; shift; shift
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for i in $(seq 1 $N); do
  env NSGPU_GRAPH=device NSGPU_THREADS=$T "$@" timeout 300 python3 bench.py --steps 1 --warmup 0 $LEAN > gpurun_out/r06_stress_$i.json 2> gpurun_out/r06_stress_$i.log
  python3 -c "import json; j=json.load(open('gpurun_out/r06_stress_$i.json')); print($i, j['value'], j['ms_per_step'], 'bad reads', j['config']['lossless_roundtrip_bad_reads'], 'parity', j.get('parity',{}).get('all_identical'))"
done
