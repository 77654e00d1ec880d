# merge policy: 512 images per pass at 64x64 (eager), hipGraph replay only for passes <= 256 images: search / bench tests, candidate default line
set -o pipefail
O=gpurun_out/r03_mergepol
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_hip_search.py tests/test_bench_multi.py tests/test_search_cli_multi.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python bench.py --workload candidate --steps 3 > $O/bench_candidate_default.json.log 2> $O/bench.err || exit 1
python bench.py --workload candidate --steps 2 --merge-batches 2 --no-cpu-baseline > $O/bench_candidate_merge2_graph.json.log 2>> $O/bench.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_mergepol/bench_*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('images_per_sec'), d['config'].get('launch'), d['config'].get('batches_per_pass'), d['time_split_s']['sample_time'], d['time_split_s']['fid_time'])
PY
