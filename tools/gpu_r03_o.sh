#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03o
export GPU_MAX_HW_QUEUES=8
NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 > gpurun_out/r03o/b.json 2> gpurun_out/r03o/b.err
grep "alignments per batch" gpurun_out/r03o/b.err
