set -o pipefail
O=gpurun_out/r03d
mkdir -p $O
python -m pytest tests/test_hip_sd.py tests/test_hip_classifier.py tests/test_hip_kernels.py -m gpu -q -k "geglu or cond_fn or linear or sd_unet or sampler" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
ADM_SD_FUSE_GEGLU=0 python bench.py --workload sd --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/sd_geglu_off.json.log 2>$O/sd_off.err
ADM_SD_FUSE_GEGLU=1 python bench.py --workload sd --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/sd_geglu_on.json.log 2>$O/sd_on.err
ADM_SD_FUSE_GEGLU=0 python bench.py --workload sd --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/sd_geglu_off2.json.log 2>>$O/sd_off.err
ADM_SD_FUSE_GEGLU=1 python bench.py --workload sd --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/sd_geglu_on2.json.log 2>>$O/sd_on.err
for f in off on off2 on2; do python -c "import json,sys; d=json.loads([l for l in open('$O/sd_geglu_$f.json.log') if l.startswith('{')][0]); print('$f', d['value'], d['ms_per_step'])"; done
export ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so
PRE=20 VARIANT=0 python tools/conv_timing.py > $O/conv_tile_timing_256cu.log 2>&1
PRE=20 VARIANT=8 python tools/conv_timing.py > $O/conv_tile_timing_256cu_variant8.log 2>&1
ADM_CONV_MAX_BLOCKS=128 PRE=20 python tools/conv_timing.py > $O/conv_tile_timing_128cu.log 2>&1
ADM_CONV_MAX_BLOCKS=32 PRE=20 python tools/conv_timing.py > $O/conv_tile_timing_32cu.log 2>&1
unset ADM_HIP_LIB
cut -c1-330 $O/conv_tile_timing_256cu.log $O/conv_tile_timing_256cu_variant8.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_adm128 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload adm128 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_adm128_profiled.json.log 2> $GRAFT_REPO_ROOT/$O/bench_adm128_profiled.err
cd $GRAFT_REPO_ROOT
find $O/stats_adm128 -name '*kernel_stats.csv' -exec cp {} $O/bench_adm128_kernel_stats.csv \; ; rm -rf $O/stats_adm128
head -12 $O/bench_adm128_kernel_stats.csv | cut -c1-160
