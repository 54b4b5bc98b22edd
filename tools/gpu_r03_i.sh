#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03i
export GPU_MAX_HW_QUEUES=8
timeout 1800 python -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "cfg1 or lockstep" 2>&1 | tail -3
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NSGPU_KSW_NO_EARLY_EXIT=1; else unset NSGPU_KSW_NO_EARLY_EXIT; fi
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 > gpurun_out/r03i/ab_$v.json 2> gpurun_out/r03i/ab_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03i/ab_$v.json")); c=d["config"]
print("no_early_exit=$v:", d["value"], "Mb/s", d["ms_per_step"], "ms; B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "dp avg launch ms", d["roofline"]["avg_launch_ms"])
PY
  grep "wait for the DP" gpurun_out/r03i/ab_$v.err | tail -1 | cut -c1-120
done
