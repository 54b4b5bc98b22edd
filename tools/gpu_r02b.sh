cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 600 python tools/check_ksw_reg.py 640 7 > gpurun_out/r02b/check.txt 2>&1
tail -30 gpurun_out/r02b/check.txt
timeout 200 python tools/bench_ksw.py 40000 > gpurun_out/r02b/ksw_short.txt 2>&1; tail -2 gpurun_out/r02b/ksw_short.txt
timeout 200 python tools/bench_ksw.py --long 1200 > gpurun_out/r02b/ksw_long.txt 2>&1; tail -2 gpurun_out/r02b/ksw_long.txt
