#!/bin/bash
# (record of a round-4 measurement run: G16_DEFER_L_RED / G16_BATCH_REDUCE were knobs of the library AT THAT COMMIT and were
# removed once measured -- profiles/r04_defer_l_reduction_ab.txt, profiles/r04_proj_k24_knob_sweep*.json)
# round 4, GPU run 1: (a) parity subset on the new sort-entry layout / memory plan / self-test,
# (b) fqmul variants microbenchmark, (c) same-box A/B of the deferred L reduction, (d) clock trace with
# HBM gathers vs cache-resident gathers (-DG16_DEBUG_GATHER variant), (e) 2^26 constraints on one GPU.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp
echo "== (a) parity subset"
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q --durations=6 \
  -k "msm_vs_oracle or hot_bucket or prove_synthetic or self_test or fewer_planes or closed_form or (headline and 20) or (in_library_multi_device_prover_large and 14) or test_in_library_multi_device_prover" \
  > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_subset.log
echo "== (b) fqmul variants"
./scripts/ubench/fqmul_variants | tee $O/fqmul_variants.txt
echo "== (c) deferred L reduction, same box"
for rep in 1 2; do for v in 0 1 2; do
  G16_DEFER_L_RED=$v G16_BENCH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 3 --cpu-log2 0 > $O/defer_${v}_$rep.json 2> $O/defer_${v}_$rep.err
  python - $O/defer_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("G16_DEFER_L_RED=%s" % sys.argv[2], round(d["ms_per_step"], 3), "ms", d["parity"], {k: round(v, 2) for k, v in d["stages_ms_per_step"].items()})
PY
done; done
echo "== (d) clock trace: HBM gathers vs cache-resident gathers"
scripts/clock_sample.sh hbm env G16_BENCH_NO_PIPELINE=1 python bench.py --steps 400 --warmup 2 --cpu-log2 0 > $O/clock_hbm.txt 2> /dev/null
scripts/clock_sample.sh cached env G16_AMD_LIB=scripts/variants/libg16_dbg_gather.so G16_DEBUG_GATHER_MASK=0x3ff G16_BENCH_NO_PIPELINE=1 \
  python bench.py --steps 400 --warmup 2 --cpu-log2 0 > $O/clock_cached.txt 2> /dev/null
python - $O <<'PY'
import json, re, sys
for name in ("hbm", "cached"):
    clk, pw, ms = [], [], None
    for l in open(f"{sys.argv[1]}/clock_{name}.txt"):
        m = re.search(r"\((\d+)Mhz\)", l)
        if m and "sclk" in l:
            clk.append(int(m.group(1)))
        m = re.search(r"Power \(W\): ([\d.]+)", l)
        if m:
            pw.append(float(m.group(1)))
        if l.startswith("{"):
            ms = json.loads(l)["ms_per_step"]
    busy = [(c, p) for c, p in zip(clk, pw) if p > 700]
    if busy:
        print(name, "busy samples", len(busy), "mean sclk %.0f MHz" % (sum(c for c, _ in busy) / len(busy)),
              "mean power %.0f W" % (sum(p for _, p in busy) / len(busy)), "ms/proof", ms)
    else:
        print(name, "no busy samples", len(clk), ms)
PY
echo "== (e) 2^26 constraints on one GPU"
free -g | head -2
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "$avail" -lt 170 ]; then echo "less than 170 GB of host memory available: not attempting 2^26"; exit 0; fi
G16_BENCH_NO_PIPELINE=1 timeout 1500 python bench.py --log2 26 --steps 2 --warmup 1 --cpu-own > $O/r04_bench_chain26.json 2> $O/chain26.err
echo "rc=$?"; tail -5 $O/chain26.err
python - $O/r04_bench_chain26.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "M/s", d.get("parity"), d["config"].get("msm"), d["stages_ms_per_step"], d.get("cpu_baseline", {}).get("samples_s"))
except Exception as e:
    print("no line", e)
PY
