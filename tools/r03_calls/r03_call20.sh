set -o pipefail
O=gpurun_out/r03s
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_driver_like.json.log 2> $O/bench.err; echo "bench rc $?"; python -c "import json; d=json.loads([l for l in open('$O/bench_driver_like.json.log') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['isolated']['frac'], d['cpu_baseline']['value'])"
