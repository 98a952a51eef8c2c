#!/bin/bash
# round 6, GPU call 12: experiment -- the batched A | B1 reduction with 65536 threads per MSM (G16_RED_PER_MSM=1)
O=gpurun_out/r6_12; mkdir -p $O; export TMPDIR=/tmp
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
for w in "c22:--steps 15 --warmup 3" "c21:--log2 21 --steps 20 --warmup 3" "c24:--log2 24 --steps 5 --warmup 1"; do
  n=${w%%:*}; a=${w#*:}
  for rep in 1 2 3; do
  for k in 0 1; do
    G16_RED_PER_MSM=$k G16_BENCH_NO_PIPELINE=1 python bench.py $a --no-pmc --cpu-log2 14 --no-secondary > $O/${n}_pm${k}_${rep}.json 2> $O/err.txt; line $O/${n}_pm${k}_${rep}.json "$n red_per_msm=$k"
  done
  done
done
