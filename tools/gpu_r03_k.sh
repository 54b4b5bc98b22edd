#!/bin/bash
# cfg3 at size: ~1 Gbase over a 4.6 Mb genome (~217x), default schedule and the 1024-builder schedule; host memory and time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03k
export GPU_MAX_HW_QUEUES=8
for cfg in "80 1 3 5" "1024 4 0 1"; do
  set -- $cfg
  NSGPU_CONS_DEBUG=1 timeout 1500 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --reads 125000 --depth 217.4 --builders $1 --groups $2 --seed-depth $3 --seed-rings $4 > gpurun_out/r03k/cfg3_$1.json 2> gpurun_out/r03k/cfg3_$1.err
  echo "rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03k/cfg3_$1.json")); c=d["config"]
    print("cfg3 B=$1:", d["value"], "Mb/s", d["ms_per_step"], "ms; contigs", c["contigs"], "lone", c["lone_reads"], "B/base", c["stream_bytes_per_base"], "rounds", c["rounds"], "bad", c["lossless_roundtrip_bad_reads"], "rss", c["host_peak_rss_gb"], "bases", c["bases_per_gpu"])
except Exception as e:
    print("FAILED", e)
PY
  tail -2 gpurun_out/r03k/cfg3_$1.err | cut -c1-200
done
