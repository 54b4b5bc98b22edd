#!/bin/bash
# repeats-genome test + bench variant, rocprofv3 kernel stats of the default bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03g
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "repeats_genome" 2>&1 | tail -3
timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --genome repeats > gpurun_out/r03g/bench_repeats.json 2> gpurun_out/r03g/bench_repeats.err
timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --genome repeats --builders 1024 --groups 4 --seed-depth 0 > gpurun_out/r03g/bench_repeats_1024.json 2> gpurun_out/r03g/bench_repeats_1024.err
python3 - <<'PY'
import json
for n in ("bench_repeats", "bench_repeats_1024"):
    try:
        d=json.load(open("gpurun_out/r03g/%s.json" % n)); c=d["config"]
        print(n, d["value"], "Mb/s; seed_pairs", c["seed_pairs"], "contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "bad", c["lossless_roundtrip_bad_reads"])
    except Exception as e:
        print(n, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03g/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --cpu-sample 0 --throughput-leg 0 > $GRAFT_REPO_ROOT/gpurun_out/r03g/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03g/bench_prof.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r03g/prof/b_kernel_trace.csv gpurun_out/r03g/prof/*/b_kernel_trace.csv
find gpurun_out/r03g/prof -name "*kernel_stats.csv" | head
python3 - <<'PY'
import csv,re,json,glob
f=glob.glob('gpurun_out/r03g/prof/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r['Calls']) for r in rows)
print("GPU operations (kernels) over 2 steps:", tot)
for r in rows[:10]:
    m=re.search(r'(\w+(<[^>(]*>)?)\(', r['Name']); n=m.group(1) if m else r['Name'][:40]
    print("%-40s calls %6s total %8.1f ms avg %7.3f ms %5s%%"%(n[:40], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
d=json.load(open('gpurun_out/r03g/bench_prof.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])
PY
