cd /root/repo
mkdir -p gpurun_out/bt
timeout 1500 python -m pytest tests/test_ksw2_gpu.py tests/test_align_gpu.py -m gpu -x -q > gpurun_out/bt/tests.log 2>&1; echo "rc=$?" >> gpurun_out/bt/tests.log
tail -4 gpurun_out/bt/tests.log
python tools/bench_ksw.py > gpurun_out/bt/ksw_new.txt 2>&1; NSGPU_KSW_SERIAL_BACKTRACK=1 python tools/bench_ksw.py > gpurun_out/bt/ksw_old.txt 2>&1
python tools/bench_ksw.py --long > gpurun_out/bt/kswl_new.txt 2>&1; NSGPU_KSW_SERIAL_BACKTRACK=1 python tools/bench_ksw.py --long > gpurun_out/bt/kswl_old.txt 2>&1
tail -1 gpurun_out/bt/ksw_new.txt gpurun_out/bt/ksw_old.txt gpurun_out/bt/kswl_new.txt gpurun_out/bt/kswl_old.txt
bash tools/gpu_ab_env.sh NSGPU_KSW_SERIAL_BACKTRACK 2
