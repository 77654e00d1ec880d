set -o pipefail
O=gpurun_out/r03z3
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_kernels.py tests/test_hip_switches.py -x -q -m gpu -k "folded_in or fold or sd_unet_non_default" -s > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
grep "skip fold\|passed\|two launches" $O/pytest.log
for i in 1 2; do
  ADM_FOLD_SKIP=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_base_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_fold_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03z3/guided_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
