#!/bin/bash
# SQ counters of the guided workload's kernels (MFMA-pipe busy, wave stall classes, LDS bank conflicts), two --pmc passes of 8 SQ
# counters each over `bench.py --steps 1` (MI355X_MICROARCH.md "rocprofv3 PMC slots"); reduced by tools/pmc_summary.py.
#   bash tools/profile_sq.sh <outdir>      (run through gpurun from the repo root; copy the csv into profiles/)
set -o pipefail
OUT=${1:-gpurun_out/r03_sq}
mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --workload guided --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/pass$i.err || exit 1
  CSV=$(find $OUT/pass$i -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py $CSV $OUT/sq_pass$i.csv conv_kernel conv1x1r attn gn_ || exit 1
  rm -rf $OUT/pass$i
done
cat $OUT/sq_pass1.csv > $OUT/pmc_sq_counters.csv; tail -n +2 $OUT/sq_pass2.csv >> $OUT/pmc_sq_counters.csv
echo done >&2
