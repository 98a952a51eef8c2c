#!/bin/bash
# round 5, end-of-round validation B: strong-scaling projections (one rank of G alone), the N > 1 launch shapes,
# the hardware-day rehearsal on one GPU (link probe table included)
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
echo "== 4. projections"
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > $O/${TAG}_proj_k22.json 2> $O/proj22.err; echo "rc=$?"
timeout 1500 python scripts/dist_projection.py 24 8 5 points,buckets > $O/${TAG}_proj_k24.json 2> $O/proj24.err; echo "rc=$?"
python - $O/${TAG}_proj_k22.json $O/${TAG}_proj_k24.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(p, "single", round(d["single_gpu_ms"], 2))
        for k, v in d["ranks"].items():
            print("  ", k, round(v["per_rank_ms"], 2), "eff", round(v["efficiency_before_xgmi"], 3), "exposed-link eff", round(v["efficiency_if_all_link_time_exposed"], 3))
    except Exception as e:
        print(p, "no line", e)
PY
echo "== 5. N > 1 launch shapes (one GPU, gloo)"
timeout 900 python -m pytest tests/test_bench_shapes.py -m gpu -x -q > $O/${TAG}_pytest_bench_shapes.log 2>&1; echo "rc=$?"; tail -3 $O/${TAG}_pytest_bench_shapes.log
echo "== 6. hardware-day rehearsal"
G16_HWDAY_FAKE=1 timeout 1500 bash scripts/hardware_day.sh > $O/${TAG}_hardware_day_rehearsal_one_gpu.txt 2>&1; echo "rc=$?"; tail -30 $O/${TAG}_hardware_day_rehearsal_one_gpu.txt
