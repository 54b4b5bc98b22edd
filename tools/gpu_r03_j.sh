#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03j
export GPU_MAX_HW_QUEUES=8
timeout 1800 python -m pytest tests/test_ksw2_gpu.py -x -q -m gpu -k "latency or overhang" 2>&1 | tail -4
for v in 700 0 400 0 1000; do
  NSGPU_KSW_LATENCY_ROWS=$v NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 > gpurun_out/r03j/ab_$v.json 2> gpurun_out/r03j/ab_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03j/ab_$v.json")); c=d["config"]
print("latency_rows=$v:", d["value"], "Mb/s", d["ms_per_step"], "ms; B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "dp avg launch ms", d["roofline"]["avg_launch_ms"], "launches", d["roofline"]["launches"])
PY
  grep "wait for the DP" gpurun_out/r03j/ab_$v.err | tail -1 | cut -c1-120
done
