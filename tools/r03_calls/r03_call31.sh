# skip-connection fold: kernel parity, then the guided bench A/B (ADM_FOLD_SKIP) on one box
set -o pipefail
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "folded_in" -s > $O/pytest_fold.log 2>&1 || { tail -40 $O/pytest_fold.log; exit 1; }
grep "skip fold rel\|passed" $O/pytest_fold.log
for i in 1 2; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_base_$i.json.log 2>> $O/bench.err || exit 1
  ADM_FOLD_SKIP=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_fold_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03z/guided_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('output_check',{}).get('checksum'))
PY
