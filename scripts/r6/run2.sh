#!/bin/bash
# round 6, GPU call 2: accumulation grid sizes against the 3-waves-per-SIMD residency of the optimistic G1 kernel
# (G16_ACC_GRID: 1536 workgroups = exactly one round, 2048 = 1.33 rounds (the 2^20 default), 3072 = two rounds (the
# 2^22 default), 4608 = three), the new bench legs (secondary block, all-cores CPU column), the SpMV with 4 lanes
O=gpurun_out/r6_2; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
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
for rep in 1 2; do
for g in 2048 1536 3072; do
  G16_ACC_GRID=$g G16_BENCH_NO_PIPELINE=1 python bench.py --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/c20_g$g.json 2> $O/c20_g$g.err; line $O/c20_g$g.json "chain20 grid=$g"
done
done
for g in 2048 1536 3072; do
  G16_ACC_GRID=$g G16_BENCH_NO_PIPELINE=1 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/p20_g$g.json 2> $O/p20_g$g.err; line $O/p20_g$g.json "poseidon20 grid=$g"
done
for rep in 1 2; do
for g in 3072 1536 4608; do
  G16_ACC_GRID=$g G16_BENCH_NO_PIPELINE=1 python bench.py --steps 15 --warmup 3 --no-pmc --cpu-log2 0 > $O/c22_g$g.json 2> $O/c22_g$g.err; line $O/c22_g$g.json "chain22 grid=$g"
done
done
for g in 2048 1024 3072; do
  G16_ACC_GRID_G2=$g G16_BENCH_NO_PIPELINE=1 python bench.py --steps 15 --warmup 3 --no-pmc --cpu-log2 0 > $O/c22_g2_$g.json 2> $O/c22_g2_$g.err; line $O/c22_g2_$g.json "chain22 gridG2=$g"
done
# the default line with its new legs
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value_pcie_inclusive"], d.get("gpu_over_cpu"), d.get("gpu_over_cpu_all_cores"))
print(json.dumps(d["cpu_baseline"], indent=0)[:1500])
print(json.dumps(d.get("secondary"), indent=0)[:3000])
PY
tail -5 $O/bench_default.err
rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r06_poseidon20_kernel_stats.txt 2>&1; grep -E "spmv|calls" $O/r06_poseidon20_kernel_stats.txt | cut -c1-150
