#!/bin/bash
# (record of a round-4 measurement run: G16_DEFER_L_RED / G16_BATCH_REDUCE were knobs of the library AT THAT COMMIT and were
# removed once measured -- profiles/r04_defer_l_reduction_ab.txt, profiles/r04_proj_k24_knob_sweep*.json)
# round 4, GPU run 2: (a) parity subset on the CURRENT library (self-test, overflow path, key-generator
# pin, planes < W at 2^22), (b) fqmul variants (DPF product with opaque FP instructions),
# (c) deferred L reduction same-box A/B at 2^22, (d) one rank of 8 at 2^24: schedule knob sweep on one key
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4b; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
echo "== (a) parity subset"
timeout 1200 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q --durations=8 \
  -k "self_test or overflow_falls or sibling or key_generator_pinned or fewer_planes or hot_bucket or test_in_library_multi_device_prover" \
  > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_subset.log
echo "== (b) fqmul variants"
./scripts/ubench/fqmul_variants | tee $O/fqmul_variants.txt
echo "== (c) deferred L reduction, same box, 2^22"
for rep in 1 2; do for v in 0 1 2; do
  G16_DEFER_L_RED=$v G16_BENCH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 3 --cpu-log2 0 > $O/defer_${v}_$rep.json 2> $O/defer_${v}_$rep.err
  python - $O/defer_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("G16_DEFER_L_RED=%s" % sys.argv[2], round(d["ms_per_step"], 3), "ms", d["parity"], {k: round(v, 2) for k, v in d["stages_ms_per_step"].items()})
PY
done; done
echo "== (d) one rank of 8 at 2^24, knob sweep"
timeout 1500 python scripts/dist_projection.py 24 8 5 points \
  "offmain:G16_REDUCE_OFF_MAIN=1,G16_BATCH_REDUCE=0;deferL1:G16_DEFER_L_RED=1;deferL2:G16_DEFER_L_RED=2;c22:G16_PROJ_WINDOW_BITS=22;c19:G16_PROJ_WINDOW_BITS=19;again:G16_PROJ_NOP=1" \
  > $O/proj_k24_sweep.json 2> $O/proj_k24_sweep.err; echo "rc=$?"; tail -3 $O/proj_k24_sweep.err
python - $O/proj_k24_sweep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("single", round(d["single_gpu_ms"], 2), d["single_msm"])
for k, v in d["ranks"].items():
    print(k, "rank ms", round(v["per_rank_ms"], 2), "eff", round(v["efficiency_before_xgmi"], 3), "c/W", v["c_w"], v["W_w"], v["stages_ms_alone"])
PY
