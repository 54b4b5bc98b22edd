#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for cfg in "64 1 3 3" "64 1 3 4" "96 1 3 5" "96 1 3 6" "48 1 3 4" "64 1 2 6" "64 1 4 4"; do
  set -- $cfg
  timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --builders $1 --groups $2 --seed-depth $3 --seed-rings $4 > gpurun_out/c_sweep_$1_$2_$3_$4.json 2> gpurun_out/c_sweep_$1_$2_$3_$4.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/c_sweep_$1_$2_$3_$4.json"))
    c=d["config"]
    print("B=$1 G=$2 d=$3 r=$4:", d["value"], "Mb/s", d["ms_per_step"], "ms; contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "bad", c["lossless_roundtrip_bad_reads"])
except Exception as e:
    print("B=$1 G=$2 d=$3 r=$4: FAILED", e)
PY
done
NSGPU_KSW_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --reads 25000 --builders 16 --groups 1 --seed-depth 3 --seed-rings 5 > gpurun_out/c_kswdbg.json 2> gpurun_out/c_kswdbg.err
grep -c KSW gpurun_out/c_kswdbg.err
