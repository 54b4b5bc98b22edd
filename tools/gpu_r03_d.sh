#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "lockstep" > gpurun_out/d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/d_tests.log
tail -4 gpurun_out/d_tests.log
for cfg in "64 1 3 4" "80 1 3 5" "96 1 3 5" "64 1 3 6" "128 1 3 8"; do
  set -- $cfg
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --builders $1 --groups $2 --seed-depth $3 --seed-rings $4 > gpurun_out/d_sweep_$1_$2_$3_$4.json 2> gpurun_out/d_sweep_$1_$2_$3_$4.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/d_sweep_$1_$2_$3_$4.json"))
    c=d["config"]
    print("B=$1 G=$2 d=$3 r=$4:", d["value"], "Mb/s", d["ms_per_step"], "ms; contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "bad", c["lossless_roundtrip_bad_reads"])
except Exception as e:
    print("B=$1 G=$2 d=$3 r=$4: FAILED", e)
PY
done
