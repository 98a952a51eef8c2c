#!/bin/bash
# round 3, GPU run 7: bench lines for profiles/ (sweep, poseidon, dense-skewed, chain 2^20) + projections with both cuts
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/bench_sweep.py 20 > gpurun_out/r03_bench_sweep.txt 2> gpurun_out/r03_bench_sweep.err
head -9 gpurun_out/r03_bench_sweep.txt
timeout 900 python bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 > gpurun_out/r03_bench_poseidon20.json 2> gpurun_out/r03_bench_poseidon20.err
timeout 900 python bench.py --workload dense-skewed --log2 20 --steps 10 --warmup 2 > gpurun_out/r03_bench_dense_skewed20.json 2> gpurun_out/r03_bench_dense_skewed20.err
timeout 900 python bench.py --log2 20 --steps 10 --warmup 2 > gpurun_out/r03_bench_chain20.json 2> gpurun_out/r03_bench_chain20.err
python - <<'PY'
import json
for n in ("poseidon20","dense_skewed20","chain20"):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r03_bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],2), "ms", round(d["value"]/1e6,1), "M/s pcie", d["ms_per_step_pcie_inclusive"], "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), d["parity"])
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > gpurun_out/r03_proj_k22.json 2> gpurun_out/r03_proj_k22.err
timeout 1500 python scripts/dist_projection.py 24 8 3 points,buckets > gpurun_out/r03_proj_k24.json 2> gpurun_out/r03_proj_k24.err
python - <<'PY'
import json
for f in ("r03_proj_k22","r03_proj_k24"):
    d=json.load(open('/root/repo/gpurun_out/%s.json'%f))
    print(f, round(d["single_gpu_ms"],2))
    for k,v in d["ranks"].items():
        print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), "w/link", round(v["efficiency_if_all_link_time_exposed"],3), v["ranks_timed"])
PY
