set -o pipefail
O=gpurun_out/r03_small1x1c
mkdir -p $O
for i in 1 2; do
  ADM_CONV_NO_SMALL1X1=1 python bench.py --workload adm256 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm256_off_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload adm256 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm256_on_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_small1x1c/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
