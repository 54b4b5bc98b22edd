#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03m
export GPU_MAX_HW_QUEUES=8
NSGPU_SKETCH_CHECK=1 timeout 1800 python -m pytest tests/test_consensus_gpu.py tests/test_align_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "not cfg2_full and not cfg3_at_size" 2>&1 | tail -3
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NSGPU_NO_RESIDENT_LISTS=1; else unset NSGPU_NO_RESIDENT_LISTS; fi
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 > gpurun_out/r03m/ab_$v.json 2> gpurun_out/r03m/ab_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03m/ab_$v.json")); c=d["config"]
print("no_resident=$v:", d["value"], "Mb/s", d["ms_per_step"], "ms; B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "contigs", c["contigs"])
PY
  grep "batches wall-ms\|index + seeds" gpurun_out/r03m/ab_$v.err | tail -2 | cut -c1-200
done
unset NSGPU_NO_RESIDENT_LISTS
timeout 600 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --builders 1024 --groups 4 --seed-depth 0 > gpurun_out/r03m/t1024.json 2> gpurun_out/r03m/t1024.err; python -c "
import json; d=json.load(open('gpurun_out/r03m/t1024.json')); print('1024/G4:', d['value'], d['config']['stream_bytes_per_base'], d['config']['contigs'])"
