set -o pipefail
O=gpurun_out/r03o
mkdir -p $O
python -m pytest tests/test_hip_search.py tests/test_bench_multi.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --workload candidate --steps 2 --no-cpu-baseline --merge-batches 1 > $O/cand_merge1.json.log 2> $O/cand1.err
python bench.py --workload candidate --steps 2 --no-cpu-baseline > $O/cand_auto.json.log 2> $O/cand2.err
python bench.py --workload candidate --steps 2 --no-cpu-baseline --merge-batches 1 > $O/cand_merge1_b.json.log 2>> $O/cand1.err
python bench.py --workload candidate --steps 2 --no-cpu-baseline > $O/cand_auto_b.json.log 2>> $O/cand2.err
for f in cand_merge1 cand_auto cand_merge1_b cand_auto_b; do python -c "import json; d=json.loads([l for l in open('$O/$f.json.log') if l.startswith('{')][0]); print('$f', d['value'], d['ms_per_step'], d['images_per_sec'], d['time_split_s']['sample_time'], d['time_split_s']['fid_time'], d['fid_values'])"; done
