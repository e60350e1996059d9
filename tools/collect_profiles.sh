#!/bin/bash
# Collects the measurements kept under profiles/ (run on the GPU box from the repo root: gpurun -- 'bash tools/collect_profiles.sh r02').
# (needs tools/build/{barrier_lab,coherent_lab} and tools/build/libmgcr_hip_timing.so = the library with gcr_resident.hip built -DMGCR_RES_TIMING, see profiles/README.md)
# rocprofv3 is given the python interpreter itself (no env / bash -c hops) and counters are collected in their own passes.
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace -d $out/kt -o kt -- python3 $B > $out/kt.log 2>&1 && python tools/rocpd_stats.py $out/kt/kt_results.db > $out/${tag}_bench_kernel_stats.csv \
 && python tools/roofline_table.py $out/${tag}_bench_kernel_stats.csv > $out/${tag}_kernel_roofline_table.md && echo "kernel trace done" \
 && rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $B > $out/pf.log 2>&1 && echo "fetch pass done" \
 && rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $B > $out/pw.log 2>&1 && echo "write pass done" \
 && python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/${tag}_pmc_hbm_traffic.csv $out/pmc_traffic.json > $out/pmc.log 2>&1 \
 && cp $out/pmc_traffic.json profiles/pmc_traffic.json \
 && rocprofv3 --kernel-trace -d $out/vc -o vc -- python3 tools/vcycle_prof.py > $out/vc.log 2>&1 && python tools/vcycle_breakdown.py $out/vc/vc_results.db > $out/${tag}_vcycle_breakdown.md && echo "vcycle done" \
 && rocprofv3 --kernel-trace -d $out/p256 -o p -- python3 bench.py --workload poisson256_gcr > $out/p256.log 2>&1 && python tools/rocpd_stats.py $out/p256/p_results.db > $out/${tag}_poisson256_kernel_stats.csv \
 && python tools/roofline_table.py $out/${tag}_poisson256_kernel_stats.csv 256 > $out/${tag}_poisson256_roofline_table.md && echo "poisson256 done" \
 && rocprofv3 --kernel-trace -d $out/ell -o e -- python3 bench.py --workload ell_slab_spmv128 > $out/ell.log 2>&1 && python tools/rocpd_stats.py $out/ell/e_results.db > $out/${tag}_ell_slab_kernel_stats.csv && echo "ell done" \
 && python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/bench.err && echo "bench done" \
 && timeout -k 10 120 tools/build/barrier_lab > $out/${tag}_barrier_lab.txt 2>&1 && timeout -k 10 120 tools/build/coherent_lab > $out/${tag}_coherent_lab.txt 2>&1 && echo "labs done" \
 && python tools/resident_timing.py > $out/${tag}_resident_timing.txt 2>&1 \
 && cp mgpreconditionedgcr_amd/libmgcr_hip.so $out/lib_keep.so && cp tools/build/libmgcr_hip_timing.so mgpreconditionedgcr_amd/libmgcr_hip.so \
 && (MGCR_RES_TIMING=1 python tools/resident_timing.py 2>&1 | grep "^resident solve" | awk 'NR%6==1' >> $out/${tag}_resident_timing.txt; cp $out/lib_keep.so mgpreconditionedgcr_amd/libmgcr_hip.so; rm -f $out/lib_keep.so) && echo "resident timing done"
rm -rf $out/kt $out/vc $out/p256 $out/ell $out/pmc_fetch $out/pmc_write
ls -la $out
