#!/bin/bash
# round 6, GPU call 8: sort-first at 2^23 / 2^24 / 2^25 (the witness map is what the H MSM waits for at large sizes),
# the SpMV with terms paired through mul2 (tests + Poseidon trace)
O=gpurun_out/r6_8; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "row_classes or huge_column or witness_map or poseidon or key_generator and dense or libsnark or more_wires" > $O/pytest_spmv.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_spmv.log
rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/ktp.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r06_poseidon20_kernel_stats.txt 2>&1; grep -E "spmv|calls" $O/r06_poseidon20_kernel_stats.txt | cut -c1-150
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("stages_ms_per_step", {})
    print(sys.argv[2], round(d["ms_per_step"], 3), "ms", d.get("clock_mhz"), d["parity"].get("proof_verifies"), {k: round(v, 2) for k, v in s.items() if v})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for rep in 1 2; do
for k in 1 0; do
  G16_SCHED_R5=$k G16_BENCH_NO_PIPELINE=1 python bench.py --workload poseidon --log2 20 --steps 30 --warmup 3 --no-pmc --cpu-log2 0 > $O/p20_r5${k}_$rep.json 2> $O/err.txt; line $O/p20_r5${k}_$rep.json "p20 sched_r5=$k"
done
done
for L in 23 24 25; do
for rep in 1 2; do
for k in 1 0; do
  G16_SCHED_R5=$k G16_BENCH_NO_PIPELINE=1 python bench.py --log2 $L --steps 5 --warmup 1 --no-pmc --cpu-log2 0 > $O/c${L}_r5${k}_$rep.json 2> $O/err.txt; line $O/c${L}_r5${k}_$rep.json "chain$L sched_r5=$k"
done
done
done
