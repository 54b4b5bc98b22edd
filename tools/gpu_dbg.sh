cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
NSGPU_CONS_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 2> gpurun_out/dbg/dbg.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
grep -v "host threads bound\|amdgpu.ids" gpurun_out/dbg/dbg.txt | tail -21
