#!/bin/bash
# HBM-side bytes per dispatch of the V-cycle's kernels (tools/vcycle_prof.py: 256^3, 6 cycles): gpurun -- 'bash tools/pmc_vcycle.sh r03'
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE -d $out/vcf --output-format csv -- python3 tools/vcycle_prof.py > $out/vcf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE -d $out/vcw --output-format csv -- python3 tools/vcycle_prof.py > $out/vcw.log 2>&1 \
 && python tools/pmc_kernels.py $out/vcf $out/vcw 5 > $out/${tag}_vcycle_pmc_kernels.md && cat $out/${tag}_vcycle_pmc_kernels.md
rm -rf $out/vcf $out/vcw
