cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02m
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02m/prof -o b -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --reads 30000 > $R/gpurun_out/r02m/bench_prof.json 2> $R/gpurun_out/r02m/bench_prof.err
python3 - <<'PY'
import csv,re,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02m/prof/b_kernel_stats.csv')))
for r in rows[:12]:
    m=re.search(r'(\w+(<[^>(]*>)?)\(', r['Name'])
    n=m.group(1) if m else r['Name'][:50]
    print("%-45s calls %6s total %8.1f ms avg %8.3f ms  %5s%%"%(n[:45], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
PY
