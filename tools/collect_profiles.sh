#!/bin/bash
# Collects the measurements kept under profiles/ (run on the GPU box from the repo root: gpurun -- 'bash tools/collect_profiles.sh r03').
# (needs tools/build/{barrier_lab,coherent_lab} and tools/build/libmgcr_hip_timing.so = the library with gcr_resident.hip built -DMGCR_RES_TIMING, see profiles/README.md)
# rocprofv3 is given the python interpreter itself (no env / bash -c hops) and counters are collected in their own passes.
set -o pipefail
tag=${1:-r03}
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
 && rocprofv3 --kernel-trace -d $out/p512 -o p -- python3 bench.py --workload poisson512_gcr > $out/p512.log 2>&1 && python tools/rocpd_stats.py $out/p512/p_results.db > $out/${tag}_poisson512_kernel_stats.csv \
 && python tools/roofline_table.py $out/${tag}_poisson512_kernel_stats.csv 512 > $out/${tag}_poisson512_roofline_table.md && echo "poisson512 done" \
 && rocprofv3 --kernel-trace -d $out/ell -o e -- python3 bench.py --workload ell_slab_spmv128 > $out/ell.log 2>&1 && python tools/rocpd_stats.py $out/ell/e_results.db > $out/${tag}_ell_slab_kernel_stats.csv && echo "ell done" \
 && python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/bench.err && echo "bench done" \
 && rocprofv3 --kernel-trace -d $out/gen -o g -- python3 bench.py --workload poisson128_gcr_general > $out/gen.log 2>&1 && python tools/rocpd_stats.py $out/gen/g_results.db > $out/${tag}_poisson128_general_kernel_stats.csv && echo "general storage done" \
 && rocprofv3 --kernel-trace -d $out/bmg -o b -- python3 bench.py --workload bcsr_mg > $out/bmg.log 2>&1 && python tools/rocpd_stats.py $out/bmg/b_results.db > $out/${tag}_bcsr_mg_kernel_stats.csv && echo "bcsr_mg done" \
 && timeout -k 10 120 tools/build/barrier_lab > $out/${tag}_barrier_lab.txt 2>&1 && timeout -k 10 120 tools/build/coherent_lab > $out/${tag}_coherent_lab.txt 2>&1 && echo "labs done" \
 && python tools/resident_timing.py > $out/${tag}_resident_timing.txt 2>&1 \
 && cp mgpreconditionedgcr_amd/libmgcr_hip.so $out/lib_keep.so && cp tools/build/libmgcr_hip_timing.so mgpreconditionedgcr_amd/libmgcr_hip.so \
 && (MGCR_RES_TIMING=1 python tools/resident_timing.py 2>&1 | grep "^resident solve" | awk 'NR%6==1' >> $out/${tag}_resident_timing.txt; cp $out/lib_keep.so mgpreconditionedgcr_amd/libmgcr_hip.so; rm -f $out/lib_keep.so) && echo "resident timing done"
# the irregular-matrix SpMV: kernel statistics, HBM traffic and L2 request counters, scattered (+-2^17) and banded (+-900) columns
for W in 131072 900; do
  export MGCR_BENCH_IRREGULAR_WINDOW=$W
  rocprofv3 --kernel-trace -d $out/irr_kt$W -o k -- python3 bench.py --workload irregular_spmv > $out/irr_kt$W.log 2>&1 && python tools/rocpd_stats.py $out/irr_kt$W/k_results.db > $out/${tag}_irregular_w${W}_kernel_stats.csv \
  && rocprofv3 --pmc FETCH_SIZE -d $out/irr_f$W --output-format csv -- python3 bench.py --workload irregular_spmv > $out/irr_f$W.log 2>&1 \
  && rocprofv3 --pmc WRITE_SIZE -d $out/irr_w$W --output-format csv -- python3 bench.py --workload irregular_spmv > $out/irr_w$W.log 2>&1 \
  && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $out/irr_t$W --output-format csv -- python3 bench.py --workload irregular_spmv > $out/irr_t$W.log 2>&1 \
  && python tools/pmc_workload.py irregular_spmv_w$W $out/irr_f$W $out/irr_w$W profiles/pmc_traffic.json $out/irr_t$W > $out/${tag}_irregular_w${W}_pmc.json && echo "irregular $W done"
  rm -rf $out/irr_kt$W $out/irr_f$W $out/irr_w$W $out/irr_t$W
done
unset MGCR_BENCH_IRREGULAR_WINDOW
cp profiles/pmc_traffic.json $out/pmc_traffic.json
timeout -k 10 120 tools/build/gather_lab > $out/${tag}_gather_lab.txt 2>&1 && echo "gather lab done"
rm -rf $out/kt $out/vc $out/p256 $out/p512 $out/ell $out/pmc_fetch $out/pmc_write $out/gen $out/bmg
ls -la $out
