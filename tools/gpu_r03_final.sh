#!/bin/bash
# the round's record run: full GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03z
export GPU_MAX_HW_QUEUES=8
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03z/tests.log 2>&1; tail -5 gpurun_out/r03z/tests.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py --steps 3 --warmup 1 ) > gpurun_out/r03z/bench.json 2> gpurun_out/r03z/bench.err; grep -E "^step|real" gpurun_out/r03z/bench.err | tail -5
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03z/bench.json'))
print(d['value'], d['ms_per_step'], d['compression']['ratio_to_reference_tN'], d['compression']['iso_compression'], d['throughput_schedule']['value'], d['throughput_schedule']['compression']['ratio_to_reference_tN'], d['cpu_baseline']['value'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic_from_profile'])
PY
