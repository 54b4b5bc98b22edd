# environment sweep on one box: each line one bench run with the given environment
cd $GRAFT_REPO_ROOT
run() {
  ( for kv in "$@"; do export "$kv"; done
    NSGPU_CONS_DEBUG=1 timeout 400 python bench.py --steps 1 --warmup 1 --cpu-sample 0 2>gpurun_out/sweep_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_per_step']; print('%-40s' % '$*', d['value'], d['ms_per_step'])"
    grep "emission cpu-ms" gpurun_out/sweep_err.txt | tail -1 | cut -c1-200 )
}
mkdir -p gpurun_out
for rep in 1 2; do
run NSGPU_PF_TAB=16
run NSGPU_PF_TAB=32
run NSGPU_PF_TAB=64
run NSGPU_PF_TAB=128
done
