#!/bin/bash
# A/B of one environment switch (0 / 1) on the headline line (128^3), interleaved inside one box:  bash tools/ab_env_head.sh MGCR_SB_REAL
VAR=$1; shift
for kv in "$@"; do export "$kv"; done
for rep in 1 2 3; do for v in 0 1; do
env $VAR=$v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/abh_$v.json || exit 1
python - $VAR $v <<'P'
import json,sys
var,v=sys.argv[1:3]
a=json.loads(open("gpurun_out/abh_%s.json"%v).read().strip().splitlines()[-1])
print(var,v,"headline", round(a["value"]), "max", round(a["timing"]["it_per_s_max"]), "roofline", a["roofline"]["frac"], flush=True)
P
done; done
