# sanity of the rebuilt libraries (same source as the collection build): kernel tests, smoke, default bench
set -o pipefail
O=gpurun_out/r03_fin3
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_cabi.py tests/test_hip_switches.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json.log 2> $O/bench.err; echo "bench rc $?"; python -c "import json; d=json.loads([l for l in open('$O/bench_default.json.log') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
