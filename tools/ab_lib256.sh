#!/bin/bash
# A/B of library variants (tools/build/ab/lib_<name>.so) on the 256^3 workloads, interleaved inside one box:  bash tools/ab_lib256.sh base t4
L=mgpreconditionedgcr_amd/libmgcr_hip.so
cp $L /tmp/lib_keep.so
for rep in 1 2; do for v in "$@"; do
cp tools/build/ab/lib_$v.so $L || exit 1
python bench.py --workload poisson256_gcr > gpurun_out/ab256_$v.json || exit 1
python bench.py --workload mg256 > gpurun_out/abmg_$v.json || exit 1
python - $v <<'P'
import json,sys
v=sys.argv[1]
c=json.loads(open("gpurun_out/ab256_%s.json"%v).read().strip().splitlines()[-1])
m=json.loads(open("gpurun_out/abmg_%s.json"%v).read().strip().splitlines()[-1])
print(v,"256^3", round(c["it_per_s"],1), [round(c["phases"][k]["us_per_iteration"],1) for k in ("xr","apply_dots","build")], "| mg256 vcycle_ms", round(m.get("vcycle_ms"),4), "seconds_to_tol", m.get("seconds_to_tol"), "outer its", m.get("outer_iterations"), flush=True)
P
done; done
cp /tmp/lib_keep.so $L
