set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_matvec_kernels.py -x -q -m gpu > gpurun_out/t_d.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_d.log
timeout -k 10 300 python tools/bench_configs.py c2 2>&1 | tail -2
MFX_RBF_FAT=0 timeout -k 10 300 python tools/bench_configs.py c2 2>&1 | tail -2
