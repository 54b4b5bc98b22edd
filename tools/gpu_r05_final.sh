#!/bin/bash
# the round's record run: smoke, default bench, kernel-trace statistics of the bench command (the GPU suite runs on its own: pytest tests -m gpu)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05z
export GPU_MAX_HW_QUEUES=8
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py --steps 3 --warmup 1 ) > gpurun_out/r05z/bench.json 2> gpurun_out/r05z/bench.err; grep -E "^step|real" gpurun_out/r05z/bench.err | tail -5
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05z/bench.json'))
print(d['value'], d['ms_per_step'], d['compression']['ratio_to_reference_tN'], d['compression']['iso_compression'], d['throughput_schedule']['value'], d['throughput_schedule']['compression']['ratio_to_reference_tN'], d['cpu_baseline']['value'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic_from_profile'], d['reference_legal_schedule']['value'], d['nonideal']['value'], d['config']['host_waits_per_slot'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05z/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --cpu-full 0 > $GRAFT_REPO_ROOT/gpurun_out/r05z/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05z/bench_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r05z/prof -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,re,json,glob
f=glob.glob('gpurun_out/r05z/prof/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("GPU operations (kernels) over 2 steps:", sum(int(r['Calls']) for r in rows))
for r in rows[:8]:
    m=re.search(r'(\w+(<[^>(]*>)?)\(', r['Name']); n=m.group(1) if m else r['Name'][:40]
    print("%-40s calls %6s total %8.1f ms avg %7.3f ms %5s%%"%(n[:40], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
d=json.load(open('gpurun_out/r05z/bench_prof.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])
PY
