#!/bin/bash
# round 6, GPU call 11: experiment -- the main stream at the aux stream's HIGH priority (G16_MAIN_PRIO=-1): the
# reductions on the main stream then do not lose the chip to the witness map's NTT passes
O=gpurun_out/r6_11; mkdir -p $O; export TMPDIR=/tmp
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
ab() {
  for rep in 1 2; do
  for k in 0 -1; do
    G16_MAIN_PRIO=$k G16_BENCH_NO_PIPELINE=1 python bench.py $2 --no-pmc --cpu-log2 0 --no-secondary > $O/${1}_mp${k}_${rep}.json 2> $O/err.txt; line $O/${1}_mp${k}_${rep}.json "$1 main_prio=$k"
  done
  done
}
ab c22 "--steps 15 --warmup 3"
ab c20 "--log2 20 --steps 30 --warmup 3"
ab p20 "--workload poseidon --log2 20 --steps 30 --warmup 3"
ab d20 "--workload dense-skewed --log2 20 --steps 30 --warmup 3"
ab c18 "--log2 18 --steps 50 --warmup 5"
ab c24 "--log2 24 --steps 5 --warmup 1"
