#!/bin/bash
# round 6, GPU call 7: the chained-carry product (F29_CHAIN_MAD): micro-benchmark, then the variant libraries
# (all kernels / G1 accumulation only) against the product library on 2^22 and 2^20 proofs, same box
O=gpurun_out/r6_7; mkdir -p $O; export TMPDIR=/tmp
timeout 300 scripts/ubench/fqmul_chain > $O/r06_fqmul_chain.txt 2>&1; cat $O/r06_fqmul_chain.txt
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
timeout 600 env G16_AMD_LIB=$PWD/circom_compat_amd/libg16_amd_chain_g1.so python -m pytest tests/test_kernels.py -m gpu -x -q -k "msm or prove_test_zkey or prove_synthetic" 2>&1 | tail -3
for rep in 1 2 3; do
for v in product chain_g1 chain; do
  if [ $v = product ]; then L=""; else L="G16_AMD_LIB=$PWD/circom_compat_amd/libg16_amd_$v.so"; fi
  env $L G16_BENCH_NO_PIPELINE=1 python bench.py --steps 15 --warmup 3 --no-pmc --cpu-log2 15 --no-secondary > $O/c22_${v}_$rep.json 2> $O/err.txt; line $O/c22_${v}_$rep.json "chain22 $v"
done
done
for rep in 1 2; do
for v in product chain_g1 chain; do
  if [ $v = product ]; then L=""; else L="G16_AMD_LIB=$PWD/circom_compat_amd/libg16_amd_$v.so"; fi
  env $L G16_BENCH_NO_PIPELINE=1 python bench.py --log2 20 --steps 30 --warmup 3 --no-pmc --cpu-log2 15 > $O/c20_${v}_$rep.json 2> $O/err.txt; line $O/c20_${v}_$rep.json "chain20 $v"
done
done
for v in product chain; do
  if [ $v = product ]; then L=""; else L="G16_AMD_LIB=$PWD/circom_compat_amd/libg16_amd_$v.so"; fi
  env $L python bench.py --mode parts --log2 22 --steps 3 --warmup 1 --no-pmc --cpu-log2 0 > $O/parts22_$v.json 2> $O/err.txt
  python -c "
import json; d=json.loads(open('$O/parts22_$v.json').read().strip().splitlines()[-1]); print('parts22 $v', d['parts_ms'])"
done
