set -o pipefail
O=gpurun_out/r03p
mkdir -p $O
python bench.py --workload adm128 --steps 3 --no-cpu-baseline > $O/bench_adm128_b32.json.log 2> $O/err.log
python bench.py --workload adm128 --steps 2 --merge-batches 4 --no-cpu-baseline > $O/bench_adm128_merge4.json.log 2>> $O/err.log
python bench.py --workload adm128 --steps 2 --merge-batches 8 --no-cpu-baseline > $O/bench_adm128_merge8.json.log 2>> $O/err.log
for f in bench_adm128_b32 bench_adm128_merge4 bench_adm128_merge8; do python -c "import json; d=json.loads([l for l in open('$O/$f.json.log') if l.startswith('{')][0]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], d['model_tflops'], r['frac'], r['isolated']['frac'], d['config']['global_batch'])"; done
