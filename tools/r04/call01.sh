set -x
python -m pytest tests/test_hip_rccl.py tests/test_bench_multi.py "tests/test_hip_fullsize.py::test_adm256_class_conditional_dynamic_unet_matches_the_reference" tests/test_hip_bigbatch.py -x -q -m gpu > gpurun_out/r04_c01_pytest.log 2>&1 && \
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04_c01_bench.json 2> gpurun_out/r04_c01_bench.err
