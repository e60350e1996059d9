#!/bin/bash
# A/B of one environment switch (0 / 1) on the stand-alone 256^3 apply and the V-cycle:  bash tools/ab_spmv256.sh MGCR_APPLY_CARRY
VAR=$1
for rep in 1 2; do for v in 0 1; do
env $VAR=$v python bench.py --workload poisson256_gcr > gpurun_out/ab256_$v.json || exit 1
env $VAR=$v python bench.py --workload mg256 > gpurun_out/abmg_$v.json || exit 1
python - $VAR $v <<'P'
import json,sys
var,v=sys.argv[1:3]
c=json.loads(open("gpurun_out/ab256_%s.json"%v).read().strip().splitlines()[-1])
m=json.loads(open("gpurun_out/abmg_%s.json"%v).read().strip().splitlines()[-1])
s=c["spmv"]
print(var,v,"spmv cold ms", round(s["ms_cold_caches"],4), "frac", round(s["frac_hbm_peak"],3), "| read-only sweep ms", round(s["cold_caches_read_only_sweep"]["ms_cold_caches"],4), "frac", round(s["cold_caches_read_only_sweep"]["frac_hbm_peak"],3), "| vcycle_ms", round(m["vcycle_ms"],4), "seconds_to_tol", round(m["seconds_to_tol"],5), flush=True)
P
done; done
