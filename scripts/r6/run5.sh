#!/bin/bash
# round 6, GPU call 5: the two halves of the new schedule apart (G16_SORT_FIRST 0 / 1, the stream split of the
# mid-sized branch always on), the width of hidden reductions (G16_RED_LANES_HIDDEN), the default line with the
# probed all-cores CPU column
O=gpurun_out/r6_5; mkdir -p $O; export TMPDIR=/tmp
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
    G16_SORT_FIRST=$k G16_BENCH_NO_PIPELINE=1 python bench.py $2 --no-pmc --cpu-log2 0 > $O/$1_sf${k}_$rep.json 2> $O/err.txt; line $O/$1_sf${k}_$rep.json "$1 sort_first=$k"
  done
  done
}
ab c20 "--log2 20 --steps 30 --warmup 3"
ab p20 "--workload poseidon --log2 20 --steps 30 --warmup 3"
ab d20 "--workload dense-skewed --log2 20 --steps 30 --warmup 3"
ab c18 "--log2 18 --steps 50 --warmup 5"
ab c19 "--log2 19 --steps 50 --warmup 5"
ab c21 "--log2 21 --steps 20 --warmup 3"
ab c22 "--steps 15 --warmup 3"
for rep in 1 2; do
for w in "c20:--log2 20" "p20:--workload poseidon --log2 20" "d20:--workload dense-skewed --log2 20" "c18:--log2 18" "c19:--log2 19"; do
  n=${w%%:*}; a=${w#*:}
  for L in 65536 32768 16384; do
    G16_RED_LANES_HIDDEN=$L G16_BENCH_NO_PIPELINE=1 python bench.py $a --steps 30 --warmup 3 --no-pmc --cpu-log2 0 > $O/${n}_rh${L}_$rep.json 2> $O/err.txt; line $O/${n}_rh${L}_$rep.json "$n red_hidden=$L"
  done
done
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value_pcie_inclusive"], d.get("gpu_over_cpu"), d.get("gpu_over_cpu_all_cores"))
print(json.dumps(d["cpu_baseline"], indent=0)[:2500])
PY
