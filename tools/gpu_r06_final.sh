#!/bin/bash
# round 6 record runs: smoke, the default bench command (timed by the shell), kernel-trace statistics of the bench command in both graph placements
set -x
mkdir -p gpurun_out/r06z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06z/smoke.txt 2>&1; tail -1 gpurun_out/r06z/smoke.txt
t0=$(date +%s)
python3 bench.py > gpurun_out/r06z/bench.json 2> gpurun_out/r06z/bench.err
t1=$(date +%s)
echo "default bench command: $((t1 - t0)) s wall" | tee gpurun_out/r06z/bench.time
python3 -c "import json; j=json.load(open('gpurun_out/r06z/bench.json')); print(j['value'], j['ms_per_step'], {k: j['roofline'][k] for k in ('bound','achieved','frac','avg_launch_ms')}, j['cpu_baseline'])"
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for mode in host device; do
  rm -rf gpurun_out/r06z/prof_$mode
  NSGPU_GRAPH=$mode timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06z/prof_$mode -o b -- python3 bench.py --steps 2 --warmup 0 $LEAN > gpurun_out/r06z/prof_$mode.json 2> gpurun_out/r06z/prof_$mode.err
  f=$(find gpurun_out/r06z/prof_$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r06z/kernel_stats_$mode.csv
  rm -rf gpurun_out/r06z/prof_$mode
  python3 -c "import json; j=json.load(open('gpurun_out/r06z/prof_$mode.json')); print('$mode under rocprof', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done

# where the graphs should live, by host threads (one first step each), and the graph kernels' own account of a step
THREADS="2 4 6 8 16" tools/gpu_r06_threads.sh > gpurun_out/r06z/placement_by_threads.txt 2>&1
cat gpurun_out/r06z/placement_by_threads.txt
NSGPU_GRAPH=device NSGPU_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r06z/graph_kernels.json 2> gpurun_out/r06z/graph_kernels.log
grep -h "consensus graphs in HBM\|graph kernels\|one-group slot\|wall-ms: begin" gpurun_out/r06z/graph_kernels.log | tail -7 > gpurun_out/r06z/graph_kernels.txt
python3 -c "import json; j=json.load(open('gpurun_out/r06z/graph_kernels.json')); print('graphs in HBM, 16 threads:', j['value'], 'Mbases/s', j['ms_per_step'], 'ms per step, streams equal to the fixture:', j['parity']['all_identical'])" >> gpurun_out/r06z/graph_kernels.txt
cat gpurun_out/r06z/graph_kernels.txt | cut -c1-300
