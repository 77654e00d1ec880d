set -o pipefail
O=gpurun_out/r03i
mkdir -p $O
python -m pytest tests/test_variants.py tests/test_hip_switches.py tests/test_hip_fullsize.py tests/test_hip_sd.py -m gpu -q -s -k "variant or switches or schedules or three_passes or side_stream or up_resblock or fp16_torso or geglu_in" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rel |fused geglu|two-pass|variant |guidance gradient|up-ResBlock|1x1 conv rel|Error" $O/pytest.log | cut -c1-220 | tail -50
