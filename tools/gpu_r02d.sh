cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02d
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r02d/pmc_long -o l -- python3 $R/tools/bench_ksw.py --long 1200 > $R/gpurun_out/r02d/pmc_long.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH --kernel-trace --output-format csv -d $R/gpurun_out/r02d/pmc_long2 -o l -- python3 $R/tools/bench_ksw.py --long 1200 > $R/gpurun_out/r02d/pmc_long2.log 2>&1
echo done
