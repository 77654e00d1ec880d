set -o pipefail
O=gpurun_out/r03x
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_classifier.py -x -q -m gpu -k "plus_embedding or zero_insertion or pool_head" > $O/pytest_units.log 2>&1; echo "rc $?" >> $O/pytest_units.log
tail -30 $O/pytest_units.log
