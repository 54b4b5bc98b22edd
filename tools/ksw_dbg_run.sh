cd /root/repo
mkdir -p gpurun_out/kd
NSGPU_KSW_DEBUG=1 python bench.py --steps 1 --warmup 0 --throughput-leg 0 --cpu-sample 0 > gpurun_out/kd/b.json 2> gpurun_out/kd/err.txt
grep "^KSW" gpurun_out/kd/err.txt > gpurun_out/kd/ksw.txt
wc -l gpurun_out/kd/ksw.txt
