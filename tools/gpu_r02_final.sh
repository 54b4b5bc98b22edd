# the round's record run: tests, smoke, bench (default flags), rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
( time timeout 900 python bench.py ) 2>> gpurun_out/r02z/bench.time > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err; grep -E "^step" gpurun_out/r02z/bench.err | tail -3; tail -4 gpurun_out/r02z/bench.time
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02z/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r02z/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02z/bench_prof.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r02z/prof/b_kernel_trace.csv
python3 - <<'PY'
import csv,re,json
rows=list(csv.DictReader(open('gpurun_out/r02z/prof/b_kernel_stats.csv')))
tot=sum(int(r['Calls']) for r in rows)
print("GPU operations (kernels) over 3 steps:", tot)
for r in rows[:8]:
    m=re.search(r'(\w+(<[^>(]*>)?)\(', r['Name']); n=m.group(1) if m else r['Name'][:40]
    print("%-40s calls %6s total %8.1f ms avg %7.3f ms %5s%%"%(n[:40], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
d=json.load(open('gpurun_out/r02z/bench.json')); print(d['value'], d['ms_per_step'], d.get('builders_penalty'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['t1']['value'])
PY
