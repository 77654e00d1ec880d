# SD: deep staged 1x1 convs (K >= 1024) on the 192-wide tile instead of the least-padding 128-wide one
set -o pipefail
O=gpurun_out/r03y
mkdir -p $O
for i in 1 2; do
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_base_$i.json.log 2>> $O/bench.err || exit 1
  ADM_WIDE_1X1=1024 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_wide_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03y/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
