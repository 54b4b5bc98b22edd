cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -2
bash tools/gpu_ab_env.sh NSGPU_KSW_FOUR_WAVES 3
for v in X NSGPU_KSW_FOUR_WAVES; do ( export $v=1; python bench.py --steps 2 --warmup 1 --cpu-sample 0 --throughput-leg 0 --builders 1024 --groups 4 --seed-depth 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1024/G4 $v', d['value'], d['ms_per_step'])" ); done
