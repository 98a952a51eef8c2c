#!/bin/bash
# round 6, GPU call 3: in-library RCCL (g16_prove_dist), SpMV with length-sorted medium rows, the batched-affine
# arithmetic micro-benchmark, G2 grid 1024 / 2048 A/B, timelines of a 2^22 and a 2^20 proof
O=gpurun_out/r6_3; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "pytest rccl rc=$?"; tail -15 $O/pytest_rccl.log
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "row_classes or huge_column or witness_map or poseidon or link_probe or sparse_b" > $O/pytest_spmv.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_spmv.log
timeout 600 scripts/ubench/batch_affine > $O/r06_batch_affine_arithmetic.txt 2>&1; cat $O/r06_batch_affine_arithmetic.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("stages_ms_per_step", {})
    print(sys.argv[2], round(d["ms_per_step"], 3), "ms", d.get("clock_mhz"), {k: round(v, 2) for k, v in s.items() if v})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for rep in 1 2 3; do
for g in 2048 1024; do
  G16_ACC_GRID_G2=$g G16_BENCH_NO_PIPELINE=1 python bench.py --steps 15 --warmup 3 --no-pmc --cpu-log2 0 > $O/c22_g2_${g}_$rep.json 2> $O/err.txt; line $O/c22_g2_${g}_$rep.json "chain22 gridG2=$g"
done
done
for g in 2048 1024; do
  G16_ACC_GRID_G2=$g G16_BENCH_NO_PIPELINE=1 python bench.py --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/c20_g2_${g}.json 2> $O/err.txt; line $O/c20_g2_${g}.json "chain20 gridG2=$g"
done
for k in 22 20; do
rm -rf /tmp/prof_t; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o kt -- python $R/bench.py --log2 $k --steps 6 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt$k.log 2>&1
cd $R; DB=$(find /tmp/prof_t -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB 260 > $O/r06_timeline_k$k.txt 2>&1
python scripts/rocpd_stats.py $DB > $O/r06_k${k}_kernel_stats.txt 2>&1
done
rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r06_poseidon20_kernel_stats.txt 2>&1; grep -E "spmv|calls" $O/r06_poseidon20_kernel_stats.txt | cut -c1-150
