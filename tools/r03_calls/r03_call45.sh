# small tiles at 16x16 incl. fused statistics; experiment: the 3x3 convs of that level on them instead of split-K 2 (SD)
set -o pipefail
O=gpurun_out/r03_small3x3
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_sd.py tests/test_hip_switches.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
ADM_CONV_SMALL3X3=1 timeout -k 10 600 python -m pytest tests/test_hip_sd.py -x -q -m gpu > $O/pytest_small3x3.log 2>&1 || { tail -30 $O/pytest_small3x3.log; exit 1; }
tail -1 $O/pytest_small3x3.log
for i in 1 2; do
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_base_$i.json.log 2>> $O/bench.err || exit 1
  ADM_CONV_SMALL3X3=1 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_small3x3_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_small3x3/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
