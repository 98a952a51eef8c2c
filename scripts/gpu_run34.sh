# refresh of the secondary bench lines with the round's final binary
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --mode parts --log2 20 --steps 5 --cpu-log2 0 > gpurun_out/r02_bench_parts_k20.json 2>/dev/null
python bench.py --workload dense-skewed --log2 20 --steps 10 > gpurun_out/r02_bench_dense20.json 2>/dev/null
python bench.py --log2 20 --steps 10 > gpurun_out/r02_bench_chain20.json 2>/dev/null
python - <<'PY'
import json
for f in ("parts_k20", "dense20", "chain20"):
    d = json.load(open(f"gpurun_out/r02_bench_{f}.json"))
    print(f, round(d["ms_per_step"], 3), round(d["value"] / 1e6, 1), d.get("ms_per_step_pcie_inclusive"), d["parity"], d.get("parts_ms"), (d.get("cpu_baseline") or {}).get("value"))
PY
