# chaining-score kernel: parity, its share of a cfg2 step (debug breakdown) and its rocprof line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/chain
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -3
NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 2> gpurun_out/chain/dbg.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
grep "chaining scores\|sketch+index" gpurun_out/chain/dbg.txt | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/chain/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/chain/prof/b_kernel_trace.csv
grep -i "chain_forward" gpurun_out/chain/prof/b_kernel_stats.csv | cut -c1-60,200-300
