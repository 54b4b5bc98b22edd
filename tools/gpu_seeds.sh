cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seeds
timeout 1500 python -m pytest tests/test_seeds_gpu.py tests/test_chain_gpu.py tests/test_align_gpu.py tests/test_consensus_gpu.py -x -q -m gpu 2>&1 | tail -15
NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 2> gpurun_out/seeds/dbg.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
grep "chaining scores\|sketch+index\|index + seeds\|thread-CPU" gpurun_out/seeds/dbg.txt | tail -8
