set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "very_different_size" > gpurun_out/t_h.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/t_h.log
