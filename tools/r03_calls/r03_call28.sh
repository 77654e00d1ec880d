# classifier variants (pools adaptive / spatial / spatial_v2, no scale-shift, conv Downsample) + the tests around the touched kernels
set -o pipefail
O=gpurun_out/r03x
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_variants.py tests/test_hip_classifier.py tests/test_cabi.py -q -m gpu -s > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
grep -i "classifier variant\|passed\|failed\|Error\|rc " $O/pytest.log | tail -30
