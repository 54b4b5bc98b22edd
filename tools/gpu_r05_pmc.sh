#!/bin/bash
# PMC passes over the DEFAULT bench command (the iso-compression schedule): HBM traffic and issue counters of the DP kernels
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05pmc
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --cpu-full 0"
timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05pmc/fetch -o f -- python3 $ARGS > $R/gpurun_out/r05pmc/fetch.json 2> $R/gpurun_out/r05pmc/fetch.err; echo "fetch rc=$?"
timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05pmc/write -o w -- python3 $ARGS > $R/gpurun_out/r05pmc/write.json 2> $R/gpurun_out/r05pmc/write.err; echo "write rc=$?"
timeout 1500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r05pmc/issue -o x -- python3 $ARGS > $R/gpurun_out/r05pmc/issue.json 2> $R/gpurun_out/r05pmc/issue.err; echo "issue rc=$?"
cd $R
F=$(find gpurun_out/r05pmc/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r05pmc/write -name "*counter_collection.csv" | head -1); X=$(find gpurun_out/r05pmc/issue -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W > gpurun_out/r05pmc/r05_pmc_ksw_traffic.json; cat gpurun_out/r05pmc/r05_pmc_ksw_traffic.json | head -12
python3 tools/pmc_issue.py $X "bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --legal-leg 0 --nonideal-leg 0 --threads-sweep 0 --cpu-full 0 (the default schedule)" > gpurun_out/r05pmc/r05_pmc_ksw_issue.json; python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05pmc/r05_pmc_ksw_issue.json"))
for k,v in d["kernels"].items(): print(k, v["launches"], v["total_ms"], v["valu_utilisation"], v["instructions_issued_per_simd_cycle"], v["waves_per_simd"], v["wave_time_share"])
PY
rm -f $F $W $X; find gpurun_out/r05pmc -name "*kernel_trace.csv" -delete
