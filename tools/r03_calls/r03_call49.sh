# experiment: SD's 640-wide 3x3 convs at 32x32 on 128-pixel tiles
set -o pipefail
O=gpurun_out/r03_small3x3_32
mkdir -p $O
ADM_CONV_SMALL3X3_32=1 timeout -k 10 600 python -m pytest tests/test_hip_sd.py tests/test_hip_fullsize.py -x -q -m gpu -k sd > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_base_$i.json.log 2>> $O/bench.err || exit 1
  ADM_CONV_SMALL3X3_32=1 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_small32_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_small3x3_32/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
