#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/e_tests.log
tail -6 gpurun_out/e_tests.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err
tail -3 gpurun_out/e_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/e_bench.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["compression"])[:600])
print(json.dumps(d["throughput_schedule"])[:700])
print(json.dumps(d["cpu_baseline"])[:400])
PY
