#!/bin/bash
# round 6, GPU call 4: the new schedule (witness sort first, L reduction on the side stream, B's assembly right
# behind the B2 reduction) against round 5's (G16_SCHED_R5=1), same box, several sizes; reduction width knob;
# windows at 2^20; 2^24 with the G2 grid rule; then the whole GPU suite with its wall time
O=gpurun_out/r6_4; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
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
ab() { # name, bench args
  for rep in 1 2; do
  for k in 1 0; do
    G16_SCHED_R5=$k G16_BENCH_NO_PIPELINE=1 python bench.py $2 --no-pmc --cpu-log2 0 > $O/${1}_r5${k}_${rep}.json 2> $O/err.txt; line $O/${1}_r5${k}_${rep}.json "$1 sched_r5=$k"
  done
  done
}
ab c20 "--log2 20 --steps 30 --warmup 3"
ab p20 "--workload poseidon --log2 20 --steps 30 --warmup 3"
ab d20 "--workload dense-skewed --log2 20 --steps 30 --warmup 3"
ab c18 "--log2 18 --steps 50 --warmup 5"
ab c16 "--log2 16 --steps 100 --warmup 10 --tables off"
ab c22 "--steps 15 --warmup 3"
ab c21 "--log2 21 --steps 20 --warmup 3"
for L in 65536 131072 32768; do
  G16_RED_LANES=$L G16_BENCH_NO_PIPELINE=1 python bench.py --steps 15 --warmup 3 --no-pmc --cpu-log2 0 > $O/c22_red$L.json 2> $O/err.txt; line $O/c22_red$L.json "chain22 red_lanes=$L"
done
for L in 65536 131072 32768; do
  G16_RED_LANES=$L G16_BENCH_NO_PIPELINE=1 python bench.py --log2 20 --steps 30 --warmup 3 --no-pmc --cpu-log2 0 > $O/c20_red$L.json 2> $O/err.txt; line $O/c20_red$L.json "chain20 red_lanes=$L"
done
for c in 17 19 20 16; do
  G16_BENCH_NO_PIPELINE=1 python bench.py --log2 20 --window-bits $c --steps 30 --warmup 3 --no-pmc --cpu-log2 0 > $O/c20_w$c.json 2> $O/err.txt; line $O/c20_w$c.json "chain20 window=$c"
done
G16_BENCH_NO_PIPELINE=1 python bench.py --log2 24 --steps 5 --warmup 1 --no-pmc --cpu-log2 0 > $O/c24.json 2> $O/err.txt; line $O/c24.json "chain24"
G16_ACC_GRID_G2=2048 G16_BENCH_NO_PIPELINE=1 python bench.py --log2 24 --steps 5 --warmup 1 --no-pmc --cpu-log2 0 > $O/c24_g2048.json 2> $O/err.txt; line $O/c24_g2048.json "chain24 gridG2=2048"
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -3; tail -16 $O/pytest_gpu.log
