# builder-count sweep on one box (throughput vs stream size)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 1024 512 256 2048; do
  timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --builders $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('builders', c['builders'], d['value'], d['ms_per_step'], 'B/base', c['stream_bytes_per_base'], 'contigs', c['contigs'], 'lone', c['lone_reads'])"
done
done
