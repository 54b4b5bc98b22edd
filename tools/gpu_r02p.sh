cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02p
P="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/r02p/issue_short -o x -- python3 $R/tools/bench_ksw.py 40000 > $R/gpurun_out/r02p/issue_short.log 2>&1
timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/r02p/issue_long -o x -- python3 $R/tools/bench_ksw.py --long 1200 > $R/gpurun_out/r02p/issue_long.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02p/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r02p/pmc_fetch.json 2> $R/gpurun_out/r02p/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02p/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r02p/pmc_write.json 2> $R/gpurun_out/r02p/pmc_write.err
cd $R
python tools/pmc_issue.py gpurun_out/r02p/issue_short/x_counter_collection.csv > gpurun_out/r02p/issue_short.json
python tools/pmc_issue.py gpurun_out/r02p/issue_long/x_counter_collection.csv > gpurun_out/r02p/issue_long.json
python tools/pmc_traffic.py gpurun_out/r02p/pmc_fetch/f_counter_collection.csv gpurun_out/r02p/pmc_write/w_counter_collection.csv > gpurun_out/r02p/traffic.json
tail -5 gpurun_out/r02p/issue_short.log; cat gpurun_out/r02p/traffic.json | head -12
rm -rf gpurun_out/r02p/pmc_fetch gpurun_out/r02p/pmc_write
