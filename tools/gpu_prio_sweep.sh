# environment sweep on one box: each line one bench run with the given environment
cd $GRAFT_REPO_ROOT
run() {
  ( for kv in "$@"; do export "$kv"; done
    NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>gpurun_out/sweep_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_per_step']; print('%-30s' % '$*', d['value'], d['ms_per_step'])"
    grep "process CPU time" gpurun_out/sweep_err.txt | tail -1 | cut -c1-130 )
}
mkdir -p gpurun_out
for rep in 1 2 3 4; do
run NSGPU_THREADS=15
run NSGPU_THREADS=14
run NSGPU_THREADS=13
done
