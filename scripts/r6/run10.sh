#!/bin/bash
# round 6, GPU call 10: C's affine conversion divided on the host (fin_final_proj) -- parity tests, then ms per proof at
# several sizes (compare with profiles/r06_schedule_ab.txt block 4 / the final_a lines of the previous binary)
O=gpurun_out/r6_10; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "prove_test_zkey or prove_synthetic or headline_sizes and 20 or determinism or public_inputs or sibling or libsnark or sparse_b or dense_skewed_2p14" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("stages_ms_per_step", {})
    print(sys.argv[2], round(d["ms_per_step"], 3), "ms", d.get("clock_mhz"), d["parity"], {k: round(v, 2) for k, v in s.items() if v and k in ("finalize", "msm_reduce")})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for w in "c16:--log2 16 --tables off --steps 100 --warmup 10" "c18:--log2 18 --steps 50 --warmup 5" "c19:--log2 19 --steps 50 --warmup 5" "c20:--log2 20 --steps 30 --warmup 3" "p20:--workload poseidon --log2 20 --steps 30 --warmup 3" "d20:--workload dense-skewed --log2 20 --steps 30 --warmup 3" "c22:--steps 15 --warmup 3"; do
  n=${w%%:*}; a=${w#*:}
  for rep in 1 2; do
    G16_BENCH_NO_PIPELINE=1 python bench.py $a --no-pmc --cpu-log2 15 --no-secondary > $O/${n}_$rep.json 2> $O/err.txt; line $O/${n}_$rep.json "$n"
  done
done
