set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_full2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_full2.log
