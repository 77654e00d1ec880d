# candidate workload eagerly (no hipGraph replay) vs reference batches per pass
set -o pipefail
O=gpurun_out/r03_merge_eager
mkdir -p $O
for M in 2 3 4 5; do
  python bench.py --workload candidate --steps 2 --merge-batches $M --no-graph --no-cpu-baseline > $O/cand_eager_merge$M.json.log 2>> $O/bench.err || exit 1
done
python bench.py --workload candidate --steps 2 --merge-batches 2 --no-cpu-baseline > $O/cand_graph_merge2.json.log 2>> $O/bench.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_merge_eager/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('images_per_sec'))
PY
