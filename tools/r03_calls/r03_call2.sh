set -o pipefail
O=gpurun_out/r03b
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
ONLY=unet VARIANT=0 REPS=10 python tools/conv_bench.py > $O/conv_bench_v0.log 2>&1 && ONLY=unet VARIANT=8 REPS=10 python tools/conv_bench.py > $O/conv_bench_v8.log 2>&1
cat $O/conv_bench_v0.log $O/conv_bench_v8.log
