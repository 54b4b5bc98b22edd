#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03n
export GPU_MAX_HW_QUEUES=8
timeout 1800 python -m pytest tests/test_chain_gpu.py tests/test_align_gpu.py tests/test_seeds_gpu.py -x -q -m gpu 2>&1 | tail -3
NSGPU_CHAIN_NO_RING=1 timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu -k "synthetic" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "repeat" 2>&1 | tail -2
for v in 0 1; do
  if [ $v = 1 ]; then export NSGPU_CHAIN_NO_RING=1; else unset NSGPU_CHAIN_NO_RING; fi
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --genome repeats --reads 25000 --builders 20 > gpurun_out/r03n/rep_$v.json 2> gpurun_out/r03n/rep_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03n/rep_$v.json")); c=d["config"]
print("no_ring=$v:", d["value"], "Mb/s", d["ms_per_step"], "ms; seed_pairs", c["seed_pairs"], "B/base", c["stream_bytes_per_base"], "contigs", c["contigs"])
PY
  grep "chaining scores" gpurun_out/r03n/rep_$v.err | cut -c1-160
done
unset NSGPU_CHAIN_NO_RING
timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --genome repeats > gpurun_out/r03n/rep_full.json 2> gpurun_out/r03n/rep_full.err
python - <<PY
import json
d=json.load(open("gpurun_out/r03n/rep_full.json")); c=d["config"]
print("repeats full:", d["value"], "Mb/s", d["ms_per_step"], "ms; seed_pairs", c["seed_pairs"], "B/base", c["stream_bytes_per_base"], "contigs", c["contigs"], "bad", c["lossless_roundtrip_bad_reads"])
PY
