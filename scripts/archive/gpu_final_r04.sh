#!/bin/bash
# End-of-round validation on one fresh box, in the order the numbers depend on each other:
#   1. PMC passes (separate rocprofv3 --pmc runs) -> profiles-ready pmc_traffic.json stamped with the library hash
#   2. pytest -m gpu, smoke()
#   3. the driver's bench command (its line then carries the stamped traffic) + a kernel-trace summary of it
#   4. strong-scaling projections (one rank of G alone) at 2^22 and 2^24
# Usage: scripts/gpu_final.sh [round-tag, default r04].  Outputs under gpurun_out/final_<tag>/ ; copy
# what should be judged into profiles/.
TAG=${1:-r04}
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
echo "== 1. PMC passes"
bash scripts/pmc_passes.sh 22 final_$TAG/pmc > $O/pmc_passes.log 2>&1; tail -8 $O/pmc_passes.log | head -7
python scripts/pmc_traffic.py gpurun_out/final_$TAG/pmc 22 $O/pmc_traffic.json && cp $O/pmc_traffic.json profiles/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final_$TAG/pmc > $O/${TAG}_pmc_k22_accumulate.txt 2>&1
echo "== 2. pytest -m gpu, smoke"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $O/${TAG}_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -14 $O/${TAG}_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== 3. bench"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/${TAG}_bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["value_pcie_inclusive"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"], d["parity"])
PY
R=$PWD; rm -rf /tmp/prof_f; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o kt -- python $R/bench.py --steps 10 --warmup 2 --cpu-log2 0 > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_f -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/${TAG}_k22_kernel_stats.txt 2>&1; head -30 $O/${TAG}_k22_kernel_stats.txt
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
