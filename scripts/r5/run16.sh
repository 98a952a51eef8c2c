#!/bin/bash
# round 5, GPU call 16: configs[4] with the 240-row Poseidon circuit (constant S-box folded: circomlib's constraint
# count): parity on the GPU (one GPU + 8 emulated ranks, both cuts; 2^17 sharded; one-hash KAT) and the bench line
# with its own PMC passes + kernel trace
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "poseidon" > $O/${TAG}_pytest_poseidon240.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_poseidon240.log; tail -3 $O/${TAG}_pytest_poseidon240.log
timeout 900 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 5 > $O/${TAG}_bench_poseidon20.json 2> $O/p20.err; echo "rc=$?"
python - $O/${TAG}_bench_poseidon20.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s pcie", d["ms_per_step_pcie_inclusive"], "B2", round(r["avg_launch_ms"], 3), "traffic", r["traffic"], d.get("clock_mhz"), d["cpu_baseline"]["value"], d["cpu_baseline"]["samples_s"], d["parity"], d["config"]["num_constraints"], d["config"]["n_vars"])
print({k: round(v, 2) for k, v in d["stages_ms_per_step"].items() if v})
PY
R=$PWD; rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/ktp.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/${TAG}_poseidon20_kernel_stats.txt 2>&1; grep -n "k_combine_large\|k_spmv_abc\|k_bucket_accumulate<g16::Fq2\|k_bucket_count<true>\|k_bucket_scatter<true>" $O/${TAG}_poseidon20_kernel_stats.txt | cut -c1-150
