# candidate workload: reference batches per pass 2 (default: 256 // 100) vs 3 / 4 / 5
set -o pipefail
O=gpurun_out/r03_merge
mkdir -p $O
for M in 2 3 4 5; do
  python bench.py --workload candidate --steps 2 --merge-batches $M --no-cpu-baseline > $O/cand_merge$M.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_merge/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('images_per_sec'), d['fid_values'][:1])
PY
