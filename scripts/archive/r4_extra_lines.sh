#!/bin/bash
# round 4: secondary bench lines on the final binary + one rank-of-8 timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4_lines; mkdir -p $O; export TMPDIR=/tmp
line() { python - $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("cpu_baseline") or {}
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s pcie", d.get("ms_per_step_pcie_inclusive") and round(d["ms_per_step_pcie_inclusive"], 2),
          "cpu", c.get("value") and round(c["value"]), c.get("samples_s"), d["parity"], (d.get("value_pipelined") or {}).get("ms_per_proof"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
python bench.py --log2 20 --steps 20 --warmup 3 > $O/r04_bench_chain20.json 2> $O/chain20.err; line $O/r04_bench_chain20.json
python bench.py --log2 20 --mode parts --steps 10 --warmup 2 --cpu-log2 0 > $O/r04_bench_parts_k20.json 2> $O/parts20.err; line $O/r04_bench_parts_k20.json
python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 > $O/r04_bench_poseidon20.json 2> $O/pos20.err; line $O/r04_bench_poseidon20.json
python bench.py --workload dense-skewed --log2 20 --steps 20 --warmup 3 > $O/r04_bench_dense_skewed20.json 2> $O/dense20.err; line $O/r04_bench_dense_skewed20.json
python bench.py --workload complex-circuit --steps 50 --warmup 5 > $O/r04_bench_complex.json 2> $O/complex.err; line $O/r04_bench_complex.json
G16_BENCH_NO_PIPELINE=1 python bench.py --log2 24 --steps 5 --warmup 1 --cpu-log2 0 > $O/r04_bench_k24_single_gpu.json 2> $O/k24.err; line $O/r04_bench_k24_single_gpu.json
G16_BENCH_NO_PIPELINE=1 python bench.py --log2 25 --steps 3 --warmup 1 --cpu-log2 0 > $O/r04_bench_chain25.json 2> $O/k25.err; line $O/r04_bench_chain25.json
R=$PWD; cd /tmp
for cfg in "24 points 0" "22 points 0"; do
  set -- $cfg
  rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_t -o trace -- python $R/scripts/dist_rank_trace.py $1 8 $2 $3 3 > $R/$O/trace_$1_$2.log 2>&1
  db=$(find /tmp/prof_t -name "*.db" | head -1)
  python $R/scripts/rocpd_timeline.py $db 150 > $R/$O/r04_rank8_timeline_k$1_$2.txt 2>&1
  tail -3 $R/$O/trace_$1_$2.log
done
