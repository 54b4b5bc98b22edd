cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log )
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
timeout 200 python tools/bench_ksw.py 40000 > gpurun_out/r02a/ksw_short.txt 2>&1
timeout 200 python tools/bench_ksw.py --long 1200 > gpurun_out/r02a/ksw_long.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_long -o l -- python3 $GRAFT_REPO_ROOT/tools/bench_ksw.py --long 1200 > $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_long.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_short -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_ksw.py 40000 > $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_short.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r02a gpurun_out/r02a/pmc_long | head -40
tail -3 gpurun_out/r02a/pytest.log; cat gpurun_out/r02a/ksw_short.txt gpurun_out/r02a/ksw_long.txt | tail -8
