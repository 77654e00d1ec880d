# resident 1x1 kernel: Cout blocks of a pixel tile shared out over neighbouring list entries when the launch has few tiles (csplit)
set -o pipefail
O=gpurun_out/r03_csplit
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_sd.py tests/test_hip_switches.py tests/test_hip_unet.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
  ADM_C1_NO_CSPLIT=1 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_off_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_on_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_csplit/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
