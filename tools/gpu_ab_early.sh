#!/bin/bash
# interleaved A/B on one box: both parts of the results ahead of the slot's end (default) against the first part only
cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --throughput-leg 0 --legal-leg 0 --nonideal-leg 0"
for rep in 1 2; do
  for v in "" "NSGPU_EARLY_ONE_PART=1"; do
    env $v timeout 600 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v' or 'default', d['value'], d['ms_per_step'], d['config']['stream_bytes_per_base'], d['config']['rounds'])"
  done
done
