# skip loop with three activation chunks in flight: parity, tile timing
set -o pipefail
O=gpurun_out/r03z5
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "folded_in" > $O/pytest_fold.log 2>&1 || { tail -40 $O/pytest_fold.log; exit 1; }
tail -1 $O/pytest_fold.log
ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so FOLD=1 PRE=20 timeout -k 10 300 python tools/conv_timing.py > $O/conv_tile_timing_fold_ring3.log 2>&1 || exit 1
cut -c1-200 $O/conv_tile_timing_fold_ring3.log
