cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
( NSGPU_SKETCH_CHECK=1 timeout 1200 python -m pytest tests/test_consensus_gpu.py -m gpu -x -q > gpurun_out/r02n/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02n/pytest.log ); tail -6 gpurun_out/r02n/pytest.log | cut -c1-300
NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r02n/bench.json 2> gpurun_out/r02n/bench.err
grep -E "slots set by|gpu mm_sketch wall-ms:|^step|batches wall" gpurun_out/r02n/bench.err | tail -4
NSGPU_SKETCH_FULL=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r02n/bench_full.json 2> gpurun_out/r02n/bench_full.err
grep -E "^step" gpurun_out/r02n/bench_full.err | tail -2
python tools/stream_hash.py 30000 512 2>/dev/null | tail -1; NSGPU_SKETCH_FULL=1 NSGPU_SKETCH_GENERAL=1 python tools/stream_hash.py 30000 512 2>/dev/null | tail -1
