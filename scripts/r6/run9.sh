#!/bin/bash
# round 6, GPU call 9: experiment -- the witness sort on its own high-priority stream, started together with the
# witness map (G16_SORT_HIPRIO=1) against the shipped schedule, same box
O=gpurun_out/r6_9; mkdir -p $O; export TMPDIR=/tmp
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
  for k in 0 1; do
    if [ $k = 1 ]; then K="G16_SORT_HIPRIO=1"; else K="G16_DUMMY=0"; fi
    env $K G16_BENCH_NO_PIPELINE=1 python bench.py $2 --no-pmc --cpu-log2 0 --no-secondary > $O/${1}_hp${k}_${rep}.json 2> $O/err.txt; line $O/${1}_hp${k}_${rep}.json "$1 sort_hiprio=$k"
  done
  done
}
ab c22 "--steps 15 --warmup 3"
ab c20 "--log2 20 --steps 30 --warmup 3"
ab p20 "--workload poseidon --log2 20 --steps 30 --warmup 3"
ab d20 "--workload dense-skewed --log2 20 --steps 30 --warmup 3"
ab c19 "--log2 19 --steps 50 --warmup 5"
ab c21 "--log2 21 --steps 20 --warmup 3"
ab c18 "--log2 18 --steps 50 --warmup 5"
ab c24 "--log2 24 --steps 5 --warmup 1"
