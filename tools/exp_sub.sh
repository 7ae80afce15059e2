#!/bin/bash
# sub-split path: geometry sweep on C3 (k sub bits, pass-1 bits); one bench line per setting under gpurun_out/$1
out=gpurun_out/${1:-exp_sub}; mkdir -p $out
for cfg in "nosub" "k2" "k3" "k3lo8" "k4" ; do
  case $cfg in
    nosub) env="RHJ_SUB=0";;
    k2) env="RHJ_SUB=1 RHJ_SUB_K=2";;
    k3) env="RHJ_SUB=1 RHJ_SUB_K=3";;
    k3lo8) env="RHJ_SUB=1 RHJ_SUB_K=3 RHJ_SUB_LO=8";;
    k4) env="RHJ_SUB=1 RHJ_SUB_K=4";;
  esac
  env $env python bench.py --workload ${2:-c3} --steps 10 --warmup 3 --no-cpu-baseline > $out/$cfg.json 2> $out/$cfg.err || { echo "$cfg failed"; tail -5 $out/$cfg.err; }
  python - <<PY
import json
try:
    d=json.loads(open("$out/$cfg.json").read().strip().splitlines()[-1])
    k=d["kernels"]; p=k["partition"]
    print("$cfg", "ms/step %.3f"%d["ms_per_step"], "G/s %.2f"%d["value"], d["config"]["path"], "k",d["config"]["sub_bits"],"lo",d["config"]["pass1_bits"],
          "| part %.3f"%p["ms"], {x:round(v["ms"],3) for x,v in p.items() if isinstance(v,dict)}, "| join %.3f"%k["join_phase_ms"], k.get("subsplit"), "frac %.3f"%d["roofline"]["frac"], "units", d["config"]["units"], "maxb", d["config"]["max_build_side"])
except Exception as e:
    print("$cfg", "no result", e)
PY
done
