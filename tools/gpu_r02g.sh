cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 600 python tools/check_ksw_reg.py 640 11 > gpurun_out/r02g/check.txt 2>&1; tail -3 gpurun_out/r02g/check.txt
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02g/bench.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
