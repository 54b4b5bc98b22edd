cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
( NSGPU_SKETCH_DEBUG=1 timeout 900 python -m pytest tests/test_mm_sketch_gpu.py -m gpu -x -q > gpurun_out/r02l/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02l/pytest.log ); tail -25 gpurun_out/r02l/pytest.log | cut -c1-300
