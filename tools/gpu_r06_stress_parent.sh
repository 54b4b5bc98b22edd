#!/bin/bash
# the stress runs of gpu_r06_stress.sh while another process of this user holds the GPU (what a sweep child of bench.py sees)
python3 -c "
import torch, time, sys
x = torch.zeros(1 << 28, device='cuda')
torch.cuda.synchronize()
print('holder ready', flush=True)
time.sleep(${HOLD:-900})
" &
H=$!
sleep 20
tools/gpu_r06_stress.sh "$@"
kill $H
