# 1x1 split-K policy A/B on one box: off / ff2 only (>= 128 chunks) / all (>= 32 chunks)
set -o pipefail
O=gpurun_out/r03u
mkdir -p $O
for i in 1 2; do
  ADM_SD_SPLITK_1X1=0 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_off_$i.json.log 2>> $O/bench.err || exit 1
  ADM_SPLITK_1X1_MIN_CHUNKS=128 python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_ff2_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03u/sd_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
