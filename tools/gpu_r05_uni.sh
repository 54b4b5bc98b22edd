#!/bin/bash
# round 5: the wave index of the multi-wave DP kernels through readfirstlane (uniform control flow): row costs and the step, new against old library
set -x
mkdir -p gpurun_out
cp nanospring_amd/lib/libnsgpu.so /tmp/new.so
cp nanospring_amd/lib/libnsgpu_old.so /tmp/old.so
python3 -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in new old; do
cp /tmp/$v.so nanospring_amd/lib/libnsgpu.so
echo "== $v gap fills"; python3 tools/bench_ksw_rows.py 0x08 2>&1 | tail -8
echo "== $v extensions"; python3 tools/bench_ksw_rows.py 0x40 2>&1 | tail -8
done
cp /tmp/new.so nanospring_amd/lib/libnsgpu.so
bash tools/gpu_ab.sh r05_uni 2
