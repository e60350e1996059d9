#!/bin/bash
# Kernel statistics and PMC passes of the irregular-matrix SpMV workload (bench.py --workload irregular_spmv), two column windows.
#   tools/irregular_prof.sh <outdir>      (on the GPU box; rocprofv3 gets the program itself after --)
set -u
OUT=${1:-gpurun_out/irregular}
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for W in 4096 131072; do
  export MGCR_BENCH_IRREGULAR_WINDOW=$W
  python3 bench.py --workload irregular_spmv > $OUT/wl_w$W.json 2> $OUT/wl_w$W.err || echo "workload failed w=$W"
  rocprofv3 --kernel-trace --stats -d $OUT/stats_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > /dev/null 2> $OUT/stats_w$W.err || echo "stats failed"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > /dev/null 2> $OUT/fetch_w$W.err || echo "fetch failed"
  rocprofv3 --pmc WRITE_SIZE -d $OUT/write_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > /dev/null 2> $OUT/write_w$W.err || echo "write failed"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $OUT/tcc_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > /dev/null 2> $OUT/tcc_w$W.err || echo "tcc failed"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > /dev/null 2> $OUT/sq_w$W.err || echo "sq failed"
done
# keep only the small csv summaries (the merge back is capped at 64 MiB)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
du -sh $OUT
