set -o pipefail
O=gpurun_out/r03r
mkdir -p $O
B="python bench.py"
$B --workload candidate --steps 3 > $O/bench_candidate_default.json.log 2>$O/err.log
$B --workload candidate --steps 2 --merge-batches 1 --no-cpu-baseline > $O/bench_candidate_one_batch_per_pass.json.log 2>>$O/err.log
$B --workload candidate --steps 2 --merge-batches 1 --no-graph --no-cpu-baseline > $O/bench_candidate_one_batch_eager.json.log 2>>$O/err.log
$B --workload sd --steps 10 --warmup 2 --batch 12 --no-cpu-baseline > $O/bench_sd_b12.json.log 2>>$O/err.log
$B --workload sd --steps 10 --warmup 2 --batch 24 --no-cpu-baseline > $O/bench_sd_b24.json.log 2>>$O/err.log
$B --steps 3 --batch 512 --no-cpu-baseline > $O/bench_guided_b512.json.log 2>>$O/err.log
$B --steps 5 --no-cpu-baseline > $O/bench_guided_b256.json.log 2>>$O/err.log
for f in $O/bench_*.json.log; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
for ln in open(f):
    if ln.startswith('{'):
        d=json.loads(ln); r=d.get('roofline') or {}
        print(f.split('/')[-1][:44].ljust(44), d['value'], d['unit'], 'ms', d['ms_per_step'], '| roof', r.get('frac'), (r.get('isolated') or {}).get('frac'), d.get('images_per_sec'), (d.get('time_split_s') or {}).get('sample_time'))
PY
done
