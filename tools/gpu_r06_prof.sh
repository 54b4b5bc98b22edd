#!/bin/bash
# round 6: kernel-trace statistics of one lean cfg2 step with the consensus graphs in HBM and on the host (the same box, one after the other)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --graph-leg 0 --cfg3-leg 0"
for mode in device host; do
  rm -rf gpurun_out/prof_$mode
  NSGPU_GRAPH=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$mode -o p -- python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/prof_$mode.json 2> gpurun_out/prof_$mode.err
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode: $f"; head -25 "$f" | cut -c1-200
  cp "$f" gpurun_out/r06_prof_${mode}_kernel_stats.csv
  # the trace itself is large: keep only the statistics
  rm -rf gpurun_out/prof_$mode
done
