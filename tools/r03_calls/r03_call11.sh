set -o pipefail
O=gpurun_out/r03j
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -3 $O/smoke.log
python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo "bench rc $?"; python -c "import json; d=json.loads([l for l in open('$O/bench_default.json.log') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['isolated']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60], d['output_check']['finite'])"
