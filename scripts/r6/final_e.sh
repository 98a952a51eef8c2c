#!/bin/bash
# round 6, end-of-round validation E: 2^26 on one GPU with the FINAL binary (this round changed plan_msm_configs:
# the sparse-B view is budgeted in the memory plan): fewer planes than windows, pairing-verified line, no CPU proof
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
timeout 1500 python bench.py --log2 26 --steps 3 --warmup 1 --cpu-log2 0 --no-pmc --no-secondary > $O/${TAG}_bench_chain26.json 2> $O/k26.err; echo "rc=$?"
tail -3 $O/k26.err
python - $O/${TAG}_bench_chain26.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 2), "M/s", d["parity"], d["config"]["msm"], d["stages_ms_per_step"])
PY
