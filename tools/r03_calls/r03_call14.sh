set -o pipefail
O=gpurun_out/r03m
mkdir -p $O
ADM_TORSO=fp16 python -m pytest tests -q -m gpu > $O/pytest_fp16env.log 2>&1; echo "pytest rc $?" >> $O/pytest_fp16env.log; tail -12 $O/pytest_fp16env.log | cut -c1-200
