# 1x1 split-K (staged kernel over chunks [cb, ce)): parity, then the SD bench A/B on one box
set -o pipefail
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "split_k" > $O/pytest_splitk.log 2>&1 || { tail -30 $O/pytest_splitk.log; exit 1; }
tail -2 $O/pytest_splitk.log
timeout -k 10 600 python -m pytest tests/test_hip_sd.py tests/test_hip_switches.py -x -q -m gpu > $O/pytest_sd.log 2>&1 || { tail -30 $O/pytest_sd.log; exit 1; }
tail -2 $O/pytest_sd.log
for i in 1 2; do
  ADM_SD_SPLITK_1X1=0 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_off_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_on_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03t/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
