# final build (skip fold on, classifier variants, ABI 8): the whole GPU suite, smoke, and the bench lines of every workload
set -o pipefail
O=gpurun_out/r03_fold
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
B="python bench.py"
$B --steps 5 > $O/bench_guided_default.json.log 2>$O/err.log && echo guided ok
$B --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_guided_20steps.json.log 2>>$O/err.log
$B --steps 5 --with-fid --no-cpu-baseline > $O/bench_guided_with_fid.json.log 2>>$O/err.log
$B --steps 5 --torso fp16 --classifier-torso fp16 --no-cpu-baseline > $O/bench_guided_fp16_clf_fp16.json.log 2>>$O/err.log
$B --workload unguided --steps 5 --no-cpu-baseline > $O/bench_unguided.json.log 2>>$O/err.log
$B --workload adm128 --steps 3 > $O/bench_adm128_default.json.log 2>>$O/err.log && echo adm128 ok
$B --workload adm128 --steps 3 --merge-batches 1 --no-cpu-baseline > $O/bench_adm128_b32.json.log 2>>$O/err.log
$B --workload adm256 --steps 3 > $O/bench_adm256_default.json.log 2>>$O/err.log
$B --workload adm256 --skip-layers auto --with-fid --steps 3 --no-cpu-baseline > $O/bench_adm256_skip_fid.json.log 2>>$O/err.log
$B --workload sd --steps 20 --warmup 2 > $O/bench_sd_default.json.log 2>>$O/err.log && echo sd ok
$B --workload candidate --steps 3 > $O/bench_candidate_default.json.log 2>>$O/err.log && echo cand ok
for f in $O/bench_*.json.log; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
for ln in open(f):
    if ln.startswith('{'):
        d=json.loads(ln); r=d.get('roofline') or {}
        print(f.split('/')[-1][:34].ljust(34), d['value'], d['unit'], 'ms', d['ms_per_step'], '| roof', r.get('frac'), (r.get('isolated') or {}).get('frac'), r.get('avg_launch_us'), d.get('model_tflops'))
PY
done
