# stream-priority / hardware-queue sweep on one box: each line one bench run with the given environment
cd $GRAFT_REPO_ROOT
run() {
  ( for kv in "$@"; do export "$kv"; done
    timeout 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_per_step']; print('%-75s' % '$*', d['value'], d['ms_per_step'], 'index', s['consensus_index'], 'queries', s['window_queries'], 'dp_wall', s['align_dp_kernel_wall'], 'graph', s['graph_host_wall'])" )
}
for rep in 1 2 3 4; do
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=32
done
