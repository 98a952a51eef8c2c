#!/bin/bash
# round 6, GPU call 1: row-length-adaptive SpMV (spmv.h) -- parity tests, the Poseidon 2^20 kernel trace
# (k_spmv_* of the witness map and of the key generator), this round's baseline lines on the same box
O=gpurun_out/r6_1; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "row_classes or huge_column or witness_map or trapdoor_setup or libsnark or poseidon or more_wires" --durations=5 > $O/pytest_spmv.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_spmv.log
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "key_generator and (dense or chain-12 or chain-16)" > $O/pytest_keygen.log 2>&1; echo "pytest keygen rc=$?"; tail -3 $O/pytest_keygen.log
rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r06_poseidon20_kernel_stats.txt 2>&1; head -40 $O/r06_poseidon20_kernel_stats.txt | cut -c1-170
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s pcie", d.get("ms_per_step_pcie_inclusive"), d["parity"], d.get("clock_mhz"), d.get("stages_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 --no-pmc > $O/bench_poseidon20.json 2> $O/p20.err; line $O/bench_poseidon20.json
python bench.py --steps 20 --warmup 5 --no-pmc --cpu-log2 0 > $O/bench_default.json 2> $O/def.err; line $O/bench_default.json
python bench.py --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/bench_chain20.json 2> $O/c20.err; line $O/bench_chain20.json
rm -rf /tmp/prof_f; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o kt -- python $R/bench.py --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt22.log 2>&1
cd $R; DB=$(find /tmp/prof_f -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r06_k22_kernel_stats_baseline.txt 2>&1; head -30 $O/r06_k22_kernel_stats_baseline.txt | cut -c1-170
