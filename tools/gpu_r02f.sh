cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02f
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/prof -o b -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/r02f/bench_prof.json 2> $R/gpurun_out/r02f/bench_prof.err
ls $R/gpurun_out/r02f/prof | head
head -25 $R/gpurun_out/r02f/prof/b_kernel_stats.csv | cut -c1-200
