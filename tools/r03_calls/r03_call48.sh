# final build of the round: the whole GPU suite, smoke, driver-like guided line, SD / adm128 default lines
set -o pipefail
O=gpurun_out/r03_final
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_guided_final.json.log 2> $O/bench.err; echo "bench rc $?"
python bench.py --workload sd --steps 20 --warmup 2 > $O/bench_sd_final.json.log 2>> $O/bench.err; echo "sd rc $?"
python bench.py --workload adm128 --steps 3 --no-cpu-baseline > $O/bench_adm128_final.json.log 2>> $O/bench.err; echo "adm128 rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_final/bench_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['frac'], (r.get('isolated') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
PY
