set -o pipefail
O=gpurun_out/r03_csplit2
mkdir -p $O
for i in 1 2; do
  ADM_C1_NO_CSPLIT=1 python bench.py --workload adm128 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm128_off_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload adm128 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm128_on_$i.json.log 2>> $O/bench.err || exit 1
done
ADM_C1_NO_CSPLIT=1 python bench.py --workload candidate --steps 2 --no-cpu-baseline > $O/cand_off.json.log 2>> $O/bench.err || exit 1
python bench.py --workload candidate --steps 2 --no-cpu-baseline > $O/cand_on.json.log 2>> $O/bench.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_csplit2/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
