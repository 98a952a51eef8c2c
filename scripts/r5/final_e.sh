#!/bin/bash
# round 5, end-of-round validation E: the driver's command once more on another box (box-to-box spread of the
# headline and of the self-measured traffic / clock), and the configs[4] line with its own PMC passes
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default_box2.json 2> $O/bench_default_box2.err; echo "rc=$?"
timeout 900 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 5 > $O/${TAG}_bench_poseidon20_pmc.json 2> $O/p20pmc.err; echo "rc=$?"
python - $O/${TAG}_bench_default_box2.json $O/${TAG}_bench_poseidon20_pmc.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1]); r = d["roofline"]
    print(p.split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s; B2", round(r["avg_launch_ms"], 3), "ms frac", round(r["frac"], 5), "traffic", r["traffic"], r["traffic_source"][:40], d.get("clock_mhz"), d.get("power_w"), d["parity"])
PY
