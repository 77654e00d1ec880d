# skip fold on by default: the model-level parity tests, then SD / adm128 / adm256 A/B on one box
set -o pipefail
O=gpurun_out/r03z2
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_fullsize.py tests/test_hip_unet.py tests/test_hip_sd.py tests/test_hip_classifier.py tests/test_variants.py tests/test_hip_bigbatch.py -x -q -m gpu > $O/pytest_models.log 2>&1 || { tail -40 $O/pytest_models.log; exit 1; }
tail -2 $O/pytest_models.log
for W in sd adm128 adm256; do
  ADM_FOLD_SKIP=0 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_base.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_fold.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03z2/*_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
