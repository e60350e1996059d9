#!/bin/bash
# HBM-side traffic of the 256^3 GCR kernels next to their byte models: two PMC passes + a kernel trace of bench.py --workload poisson256_gcr
#   gpurun -- 'bash tools/pmc_256.sh r03'      -> gpurun_out/<tag>/<tag>_poisson256_roofline_table_pmc.md
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $out/p256k -o p -- python3 bench.py --workload poisson256_gcr > $out/p256k.log 2>&1 && python tools/rocpd_stats.py $out/p256k/p_results.db > $out/p256_stats.csv \
 && rocprofv3 --pmc FETCH_SIZE -d $out/p256f --output-format csv -- python3 bench.py --workload poisson256_gcr > $out/p256f.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE -d $out/p256w --output-format csv -- python3 bench.py --workload poisson256_gcr > $out/p256w.log 2>&1 \
 && python tools/roofline_table.py $out/p256_stats.csv 256 --pmc $out/p256f $out/p256w > $out/${tag}_poisson256_roofline_table_pmc.md && cat $out/${tag}_poisson256_roofline_table_pmc.md
rm -rf $out/p256k $out/p256f $out/p256w
