set -o pipefail
O=gpurun_out/r03v
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_switches.py tests/test_hip_sd.py tests/test_hip_kernels.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
