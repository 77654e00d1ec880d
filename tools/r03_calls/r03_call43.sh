# 128-pixel tiles for deep 1x1 convs at 16x16 as the default rule: kernel / SD / variants tests, then adm256 / adm128 / sd bench lines
set -o pipefail
O=gpurun_out/r03_small1x1b
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_sd.py tests/test_hip_bigbatch.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for W in adm256 adm128 sd; do
  python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_1.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_2.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_small1x1b/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
