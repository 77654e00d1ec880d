set -o pipefail
O=gpurun_out/r03a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 5 > $O/bench_guided_default.json.log 2> $O/bench_guided_default.err && tail -c 600 $O/bench_guided_default.json.log &&
python bench.py --workload adm128 --steps 3 > $O/bench_adm128.json.log 2> $O/bench_adm128.err && tail -c 400 $O/bench_adm128.json.log &&
python bench.py --workload candidate --steps 2 --no-cpu-baseline > $O/bench_candidate.json.log 2> $O/bench_candidate.err && tail -c 1500 $O/bench_candidate.json.log &&
python bench.py --workload adm256 --skip-layers auto --with-fid --steps 2 --no-cpu-baseline > $O/bench_adm256_skip_fid.json.log 2> $O/bench_adm256_skip_fid.err && tail -c 400 $O/bench_adm256_skip_fid.json.log
