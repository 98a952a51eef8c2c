#!/bin/bash
# round 5, end-of-round validation C: 2^26 on one GPU with this round's binary (fewer planes than windows, the
# two-level NTT tables with the round-5 renormalisation rule): pairing-verified line, no CPU proof (4 min of host time)
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python bench.py --log2 26 --steps 3 --warmup 1 --cpu-log2 0 --no-pmc > $O/${TAG}_bench_chain26.json 2> $O/k26.err; echo "rc=$?"
python - $O/${TAG}_bench_chain26.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 2), "M/s", d["parity"], d["config"]["msm"], d["stages_ms_per_step"])
PY
