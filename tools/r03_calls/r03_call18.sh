set -o pipefail
O=gpurun_out/r03q
mkdir -p $O
python -m pytest tests/test_hip_search.py tests/test_search_cli_multi.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
