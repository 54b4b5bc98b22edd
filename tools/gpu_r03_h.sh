#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03h
export GPU_MAX_HW_QUEUES=8
NSGPU_CONS_DEBUG=1 NSGPU_SKETCH_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --genome repeats --reads 25000 --builders 20 > gpurun_out/r03h/rep.json 2> gpurun_out/r03h/rep.err
grep -c "fused path gives up" gpurun_out/r03h/rep.err
grep "fused path gives up" gpurun_out/r03h/rep.err | head -3
grep "\[cons\]" gpurun_out/r03h/rep.err | grep -v "emission\|resident\|update_graph\|main path" | cut -c1-300
