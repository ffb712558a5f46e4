set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rbf or slq" > gpurun_out/t_c.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t_c.log
for i in 1 2 3; do timeout -k 10 120 python tools/bench_grad.py 131072 2560 3 2>&1 | tail -1; done
