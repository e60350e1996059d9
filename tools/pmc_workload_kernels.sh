#!/bin/bash
# HBM-side bytes per dispatch of every kernel of one bench workload:  gpurun -- 'bash tools/pmc_workload_kernels.sh poisson128_gcr_general r03'
set -o pipefail
w=$1; tag=${2:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE -d $out/wf --output-format csv -- python3 bench.py --workload $w > $out/wf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE -d $out/ww --output-format csv -- python3 bench.py --workload $w > $out/ww.log 2>&1 \
 && python tools/pmc_kernels.py $out/wf $out/ww 5 > $out/${tag}_${w}_pmc_kernels.md && cat $out/${tag}_${w}_pmc_kernels.md
rm -rf $out/wf $out/ww
