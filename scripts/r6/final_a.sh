#!/bin/bash
# round 6, end-of-round validation A (one fresh box): the GPU suite (driver's command) with its wall time, smoke, the
# driver's bench command, a kernel-trace summary of the same command, the secondary lines.
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
echo "== 1. pytest -m gpu, smoke"
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_pytest_gpu_final.log 2>&1 ) 2>&1 | tail -3; tail -18 $O/${TAG}_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== 2. bench (the driver's command)"
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - $O/${TAG}_bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["value_pcie_inclusive"], r["frac"], r.get("traffic"), r["traffic_source"][:90], d.get("clock_mhz"), d.get("power_w"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("value_all_cores"), d["cpu_baseline"]["samples_s"], d["parity"])
print(json.dumps(d.get("secondary"))[:1200])
PY
R=$PWD; rm -rf /tmp/prof_f; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o kt -- python $R/bench.py --steps 10 --warmup 2 --cpu-log2 0 --no-pmc --no-secondary > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_f -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/${TAG}_k22_kernel_stats.txt 2>&1; head -24 $O/${TAG}_k22_kernel_stats.txt | cut -c1-160
python scripts/rocpd_timeline.py $DB 260 > $O/${TAG}_timeline_k22.txt 2>&1
echo "== 3. secondary lines"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("cpu_baseline") or {}
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s pcie", d.get("ms_per_step_pcie_inclusive"), "cpu", c.get("value"), d["parity"], "tables", d["config"]["msm"].get("fixed_tables"), "sparse_b", d["config"]["msm"].get("sparse_b"), d.get("parts_ms", ""))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
python bench.py --mode parts --log2 22 --steps 5 --warmup 1 --cpu-log2 0 --no-pmc > $O/${TAG}_bench_parts_k22.json 2> $O/parts22.err; line $O/${TAG}_bench_parts_k22.json
python bench.py --mode parts --log2 20 --steps 5 --warmup 1 --cpu-log2 0 --no-pmc > $O/${TAG}_bench_parts_k20.json 2> $O/parts20.err; line $O/${TAG}_bench_parts_k20.json
python bench.py --log2 20 --steps 20 --warmup 3 --no-pmc > $O/${TAG}_bench_chain20.json 2> $O/c20.err; line $O/${TAG}_bench_chain20.json
python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 > $O/${TAG}_bench_poseidon20.json 2> $O/p20.err; line $O/${TAG}_bench_poseidon20.json
python bench.py --workload dense-skewed --log2 20 --steps 20 --warmup 3 --no-pmc > $O/${TAG}_bench_dense_skewed20.json 2> $O/d20.err; line $O/${TAG}_bench_dense_skewed20.json
python bench.py --workload complex-circuit --steps 200 --warmup 20 --no-pmc > $O/${TAG}_bench_complex.json 2> $O/cx.err; line $O/${TAG}_bench_complex.json
python scripts/bench_sweep.py 50 > $O/${TAG}_bench_sweep.txt 2> $O/sweep.err; head -9 $O/${TAG}_bench_sweep.txt
python bench.py --log2 24 --steps 5 --warmup 1 --cpu-log2 0 --no-pmc > $O/${TAG}_bench_k24_single_gpu.json 2> $O/k24.err; line $O/${TAG}_bench_k24_single_gpu.json
python bench.py --log2 25 --steps 3 --warmup 1 --cpu-log2 0 --no-pmc > $O/${TAG}_bench_chain25.json 2> $O/k25.err; line $O/${TAG}_bench_chain25.json
rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/ktp.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/${TAG}_poseidon20_kernel_stats.txt 2>&1; grep -E "spmv|calls" $O/${TAG}_poseidon20_kernel_stats.txt | cut -c1-150
python scripts/rocpd_timeline.py $DB 300 > $O/${TAG}_timeline_poseidon20.txt 2>&1
