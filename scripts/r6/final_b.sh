#!/bin/bash
# round 6, end-of-round validation B: strong-scaling projections (one rank of G alone), the opt-in 2^24 / 2^25 legs of
# the GPU suite, the N > 1 launch shapes, the hardware-day rehearsal, the PMC counter set over the accumulation launches
TAG=r06
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
echo "== 5. the opt-in legs of the GPU suite (G16_TEST_LARGE=1)"
( time G16_TEST_LARGE=1 timeout 1500 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "headline_sizes and 24 or capacity_point" --durations=3 > $O/${TAG}_pytest_gpu_large_optin.log 2>&1 ) 2>&1 | tail -3; tail -8 $O/${TAG}_pytest_gpu_large_optin.log
echo "== 6. N > 1 launch shapes (one GPU, gloo)"
timeout 900 python -m pytest tests/test_bench_shapes.py -m gpu -x -q > $O/${TAG}_pytest_bench_shapes.log 2>&1; echo "rc=$?"; tail -3 $O/${TAG}_pytest_bench_shapes.log
echo "== 7. hardware-day rehearsal"
G16_HWDAY_FAKE=1 timeout 1500 bash scripts/hardware_day.sh > $O/${TAG}_hardware_day_rehearsal_one_gpu.txt 2>&1; echo "rc=$?"; tail -12 $O/${TAG}_hardware_day_rehearsal_one_gpu.txt
echo "== 8. PMC counter set over the accumulation launches"
bash scripts/pmc_passes.sh 22 final_$TAG/pmc > $O/pmc_passes.log 2>&1; tail -9 $O/pmc_passes.log | head -8
python scripts/pmc_traffic.py gpurun_out/final_$TAG/pmc 22 $O/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final_$TAG/pmc > $O/${TAG}_pmc_k22_accumulate.txt 2>&1; head -30 $O/${TAG}_pmc_k22_accumulate.txt
