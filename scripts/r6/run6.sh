#!/bin/bash
# round 6, GPU call 6: the final rules (sort first only for large bucket sets, hidden reductions at 32768 threads, L
# reduction on the side stream) against round 5's schedule, the 8-rank projections, Poseidon kernel trace
O=gpurun_out/r6_6; mkdir -p $O; export TMPDIR=/tmp
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
ab c19 "--log2 19 --steps 50 --warmup 5"
ab c20 "--log2 20 --steps 30 --warmup 3"
ab p20 "--workload poseidon --log2 20 --steps 30 --warmup 3"
ab d20 "--workload dense-skewed --log2 20 --steps 30 --warmup 3"
ab c18 "--log2 18 --steps 50 --warmup 5"
ab c17 "--log2 17 --steps 50 --warmup 5"
ab c22 "--steps 15 --warmup 3"
ab cx "--workload complex-circuit --steps 200 --warmup 20"
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > $O/r06_proj_k22.json 2> $O/proj22.err; echo "proj22 rc=$?"
timeout 1500 python scripts/dist_projection.py 24 8 5 points,buckets > $O/r06_proj_k24.json 2> $O/proj24.err; echo "proj24 rc=$?"
python - $O/r06_proj_k22.json $O/r06_proj_k24.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], json.dumps(d)[:1500])
    except Exception as e:
        print(f, "no json:", e)
PY
