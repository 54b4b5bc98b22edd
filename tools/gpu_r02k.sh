cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
( timeout 900 python -m pytest tests/test_mm_sketch_gpu.py tests/test_align_gpu.py tests/test_consensus_gpu.py -m gpu -x -q > gpurun_out/r02k/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02k/pytest.log ); tail -5 gpurun_out/r02k/pytest.log
NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err
grep -E "slots set by|gpu mm_sketch wall-ms:|^step|batches wall" gpurun_out/r02k/bench.err | tail -8
NSGPU_SKETCH_GENERAL=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r02k/bench_gen.json 2> gpurun_out/r02k/bench_gen.err
grep -E "^step" gpurun_out/r02k/bench_gen.err | tail -3
