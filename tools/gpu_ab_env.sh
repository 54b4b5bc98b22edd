# A/B on one box of an environment switch: bash tools/gpu_ab_env.sh VAR [runs]   (prints value, ms/step, process CPU-s of the last step)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in $(seq 1 ${2:-3}); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $1=1; else unset $1; fi
    NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>gpurun_out/ab_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_per_step']; print('$1=$v', d['value'], d['ms_per_step'], 'index', s['consensus_index'], 'dp_wall', s['align_dp_kernel_wall'], 'graph', s['graph_host_wall'])"
    grep "process CPU time\|emission cpu-ms" gpurun_out/ab_err.txt | tail -2 | cut -c1-150
  done
done
