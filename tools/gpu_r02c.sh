cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02c
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/r02c/sq_counters.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_short -o s -- python3 $R/tools/bench_ksw.py 40000 > $R/gpurun_out/r02c/pmc_short.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_long -o l -- python3 $R/tools/bench_ksw.py --long 1200 > $R/gpurun_out/r02c/pmc_long.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_short2 -o s -- python3 $R/tools/bench_ksw.py 40000 > $R/gpurun_out/r02c/pmc_short2.log 2>&1
tail -3 $R/gpurun_out/r02c/pmc_short.log $R/gpurun_out/r02c/pmc_short2.log
