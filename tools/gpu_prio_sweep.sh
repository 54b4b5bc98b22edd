# environment sweep on one box: each line one bench run with the given environment
cd $GRAFT_REPO_ROOT
run() {
  ( for kv in "$@"; do export "$kv"; done
    NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>gpurun_out/sweep_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_per_step']; print('%-75s' % '$*', d['value'], d['ms_per_step'], 'index', s['consensus_index'], 'graph', s['graph_host_wall'])"
    grep "process CPU time" gpurun_out/sweep_err.txt | tail -1 )
}
mkdir -p gpurun_out
for rep in 1 2 3; do
run A=default
run MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184
run MALLOC_ARENA_MAX=4
done
