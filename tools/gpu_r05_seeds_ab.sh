#!/bin/bash
# round 5: A/B of the seeding rewrite on one box (libnsgpu_old.so = the commit before it), then cfg5's knobs against the oracle fixtures
set -x
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
cp nanospring_amd/lib/libnsgpu.so /tmp/new.so
cp nanospring_amd/lib/libnsgpu_old.so /tmp/old.so
for i in 1 2; do
  for v in new old; do
    cp /tmp/$v.so nanospring_amd/lib/libnsgpu.so
    NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_seedsab_${v}_$i.json 2> gpurun_out/r05_seedsab_${v}_$i.log
    python3 -c "import json; j=json.load(open('gpurun_out/r05_seedsab_${v}_$i.json')); print('$v', j['value'], j['ms_per_step'], j.get('parity',{}).get('all_identical'))"
    grep -E "sketch\.\.chain of lane" gpurun_out/r05_seedsab_${v}_$i.log | tail -1
  done
done
cp /tmp/new.so nanospring_amd/lib/libnsgpu.so
python3 -m pytest tests/test_consensus_gpu.py -m gpu -x -q -k "cfg5knobs" 2>&1 | tail -5
python3 tools/parity_cfg5_one_builder.py > gpurun_out/r05_parity_cfg5_one_builder.txt 2>&1
cat gpurun_out/r05_parity_cfg5_one_builder.txt
