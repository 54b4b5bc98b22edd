#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03l
export GPU_MAX_HW_QUEUES=8
NSGPU_SKETCH_CHECK=1 timeout 1800 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "lockstep or cfg1 or repeat or long_consensus or edge_cases" 2>&1 | tail -3
for i in 1 2; do
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 > gpurun_out/r03l/b_$i.json 2> gpurun_out/r03l/b_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03l/b_$i.json")); c=d["config"]
print("run $i:", d["value"], "Mb/s", d["ms_per_step"], "ms; B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "contigs", c["contigs"])
PY
  grep "batches wall-ms\|wall-ms: begin" gpurun_out/r03l/b_$i.err | tail -2 | cut -c1-200
done
