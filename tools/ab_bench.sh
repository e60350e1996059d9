#!/bin/bash
# A/B of library variants on the headline line (one-launch and three-kernel path) and the 256^3 workload:
#   bash tools/ab_bench.sh head pin2     (tools/build/ab/lib_<name>.so)
L=mgpreconditionedgcr_amd/libmgcr_hip.so
cp $L /tmp/lib_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp tools/build/ab/lib_$v.so $L || exit 1
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ab_$v.json || exit 1
  MGCR_STEPBUILD=0 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ab3_$v.json || exit 1
  python bench.py --workload poisson256_gcr > gpurun_out/ab256_$v.json || exit 1
  python - "$v" <<'P'
import json,sys
v=sys.argv[1]
a=json.loads(open("gpurun_out/ab_%s.json"%v).read().strip().splitlines()[-1])
b=json.loads(open("gpurun_out/ab3_%s.json"%v).read().strip().splitlines()[-1])
c=json.loads(open("gpurun_out/ab256_%s.json"%v).read().strip().splitlines()[-1])
print(v, "one-launch", round(a["value"]), "min", round(a["timing"]["it_per_s_max"]), "| three-kernel", round(b["value"]), "max", round(b["timing"]["it_per_s_max"]), [round(b["phases"][k]["us_per_iteration"],1) for k in ("xr","apply_dots","build")], "| 256^3", round(c["it_per_s"],1), [round(c["phases"][k]["us_per_iteration"],1) for k in ("xr","apply_dots","build")], flush=True)
P
done
done
cp /tmp/lib_keep.so $L
