#!/bin/bash
# round 6, end-of-round validation C: the driver's bench command once more on another box (the spread of the
# self-measured fields), plus smoke
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default_box2.json 2> $O/bench_default_box2.err ) 2>&1 | tail -3
python - $O/${TAG}_bench_default_box2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["ms_per_step_pcie_inclusive"], r["frac"], r.get("traffic"), d.get("clock_mhz"), d.get("power_w"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("value_all_cores"), d["cpu_baseline"]["samples_s"])
print({k: (v.get("ms_per_step") or v.get("ms_per_step_one_stream")) for k, v in d["secondary"].items()})
PY
