#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 2400 python -m pytest tests/test_dist_gpu.py tests/test_chain_gpu.py tests/test_seeds_gpu.py -x -q -m gpu > gpurun_out/f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/f_tests.log
tail -8 gpurun_out/f_tests.log
NSGPU_BENCH_BACKEND=gloo NSGPU_THREADS=8 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29877 bench.py --gpus 2 --steps 1 --warmup 0 --reads 12500 --cpu-sample 0 > gpurun_out/f_bench2.json 2> gpurun_out/f_bench2.err
tail -2 gpurun_out/f_bench2.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/f_bench2.json") if l.startswith("{")][-1])
print(d["value"], d["n_gpus"], json.dumps(d["per_rank"]), d["config"]["parallelism"][:200])
PY
