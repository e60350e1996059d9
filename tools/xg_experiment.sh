for xg in -1 1 2 4 8 16; do
  if [ $xg -lt 0 ]; then unset MGCR_XCD_GROUP; else export MGCR_XCD_GROUP=$xg; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/bx_$xg.json 2>/dev/null
done
python - <<PY
import json
for xg in (-1,1,2,4,8,16):
    d=json.load(open("gpurun_out/bx_%d.json"%xg)); print("xg",xg, round(d["value"],1), {k:round(v["us_per_iteration"],2) for k,v in d["phases"].items()})
PY
