#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03p
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu -k "tail_rings or lockstep" 2>&1 | tail -3
for cfg in "80 5 5" "80 5 3" "80 5 2" "80 5 1" "72 5 2" "72 5 1" "64 4 1"; do
  set -- $cfg
  NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --builders $1 --seed-rings $2 --seed-tail-rings $3 > gpurun_out/r03p/s_$1_$2_$3.json 2> gpurun_out/r03p/s_$1_$2_$3.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03p/s_$1_$2_$3.json")); c=d["config"]
print("B=$1 r=$2 tail=$3:", d["value"], "Mb/s", d["ms_per_step"], "ms; contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "ratio", d["compression"].get("ratio_to_reference_tN"), "rounds", c["rounds"])
PY
  grep "alignments per batch" gpurun_out/r03p/s_$1_$2_$3.err | cut -c1-170
done
