#!/bin/bash
# A/B of one environment switch (0 / 1) on the 256^3 workloads, interleaved inside one box:  bash tools/ab_env.sh MGCR_TILE_CARRY [OTHER=VALUE ...]
VAR=$1; shift
for kv in "$@"; do export "$kv"; done
for rep in 1 2; do for v in 0 1; do
env $VAR=$v python bench.py --workload poisson256_gcr > gpurun_out/ab256_$v.json || exit 1
env $VAR=$v python bench.py --workload mg256 > gpurun_out/abmg_$v.json || exit 1
python - $VAR $v <<'P'
import json,sys
var,v=sys.argv[1:3]
c=json.loads(open("gpurun_out/ab256_%s.json"%v).read().strip().splitlines()[-1])
m=json.loads(open("gpurun_out/abmg_%s.json"%v).read().strip().splitlines()[-1])
print(var,v,"256^3", round(c["it_per_s"],1), [round(c["phases"][k]["us_per_iteration"],1) for k in ("xr","apply_dots","build")], "| mg256 vcycle_ms", round(m.get("vcycle_ms"),4), "seconds_to_tol", m.get("seconds_to_tol"), "outer its", m.get("outer_iterations"), flush=True)
P
done; done
