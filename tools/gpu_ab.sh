#!/bin/bash
# A/B of two builds of the library on one box: nanospring_amd/lib/libnsgpu.so (new) against libnsgpu_old.so (built from the commit before a change),
# interleaved lean bench steps.  usage: bash tools/gpu_ab.sh <tag> [rounds] [extra bench args]
set -x
tag=${1:-ab}; rounds=${2:-2}; shift; shift
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
cp nanospring_amd/lib/libnsgpu.so /tmp/new.so
cp nanospring_amd/lib/libnsgpu_old.so /tmp/old.so
for i in $(seq 1 $rounds); do
  for v in new old; do
    cp /tmp/$v.so nanospring_amd/lib/libnsgpu.so
    NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN "$@" > gpurun_out/${tag}_${v}_$i.json 2> gpurun_out/${tag}_${v}_$i.log
    python3 -c "import json; j=json.load(open('gpurun_out/${tag}_${v}_$i.json')); print('$v', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'), j['roofline'].get('avg_launch_ms'))"
    grep -E "one-group slot" gpurun_out/${tag}_${v}_$i.log | tail -1
  done
done
cp /tmp/new.so nanospring_amd/lib/libnsgpu.so
