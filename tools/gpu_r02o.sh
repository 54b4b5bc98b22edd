cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
( timeout 900 python -m pytest tests/test_fastq.py tests/test_mm_sketch_gpu.py -m gpu -x -q > gpurun_out/r02o/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02o/pytest.log ); tail -5 gpurun_out/r02o/pytest.log | cut -c1-250
timeout 1500 python tools/backend_coders.py 100000 16 1024 256 64 > gpurun_out/r02o/backend.json 2> gpurun_out/r02o/backend.err; tail -4 gpurun_out/r02o/backend.json | cut -c1-900; tail -3 gpurun_out/r02o/backend.err
