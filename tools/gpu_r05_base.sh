#!/bin/bash
# round 5, first call: the state at the start of the round on this round's box -- slot breakdown (NSGPU_CONS_DEBUG) and the host-thread sweep
# the review asked for (what a rank gets on a shared CPU quota)
set -x
mkdir -p gpurun_out
LEAN="--throughput-leg 0 --cpu-sample 0 --cpu-full 0 --legal-leg 0 --nonideal-leg 0"
NSGPU_CONS_DEBUG=1 python3 bench.py --steps 2 --warmup 1 $LEAN > gpurun_out/r05_base_t16.json 2> gpurun_out/r05_base_t16.log
for t in 4 2; do
  NSGPU_THREADS=$t python3 bench.py --steps 1 --warmup 1 $LEAN > gpurun_out/r05_base_t$t.json 2> gpurun_out/r05_base_t$t.log
done
tail -c 600 gpurun_out/r05_base_t*.json
