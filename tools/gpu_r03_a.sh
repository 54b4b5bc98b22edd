#!/bin/bash
# round 3, first GPU call: the new parity tests, then a sweep of schedules at cfg2 (1 warm-up-free step each; --cpu-sample 0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests/test_consensus_gpu.py tests/test_fastq.py -x -q -m gpu -k "lockstep or oversize or gzip or one_builder_equals_oracle" > gpurun_out/a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
for cfg in "1024 4 0 1" "128 1 3 1" "256 1 3 1" "512 1 3 1" "256 2 3 1" "512 2 3 1" "512 4 3 1" "256 1 3 2"; do
  set -- $cfg
  timeout 600 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --builders $1 --groups $2 --seed-depth $3 --seed-rings $4 > gpurun_out/a_sweep_$1_$2_$3_$4.json 2> gpurun_out/a_sweep_$1_$2_$3_$4.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/a_sweep_$1_$2_$3_$4.json"))
    c=d["config"]
    print("B=$1 G=$2 d=$3 r=$4:", d["value"], "Mb/s", d["ms_per_step"], "ms; contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "bad", c["lossless_roundtrip_bad_reads"])
except Exception as e:
    print("B=$1 G=$2 d=$3 r=$4: FAILED", e)
PY
done
