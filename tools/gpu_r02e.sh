cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
( timeout 900 python -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py tests/test_consensus_gpu.py -m gpu -x -q > gpurun_out/r02e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e/pytest.log )
tail -4 gpurun_out/r02e/pytest.log
NSGPU_DEBUG_TIMING=1 timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
tail -5 gpurun_out/r02e/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02e/bench.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
NSGPU_KSW_NO_REG=1 timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02e/bench_noreg.json 2> gpurun_out/r02e/bench_noreg.err
python -c "
import json; d=json.load(open('gpurun_out/r02e/bench_noreg.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
