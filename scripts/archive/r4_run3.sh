#!/bin/bash
# (record of a round-4 measurement run: G16_DEFER_L_RED / G16_BATCH_REDUCE were knobs of the library AT THAT COMMIT and were
# removed once measured -- profiles/r04_defer_l_reduction_ab.txt, profiles/r04_proj_k24_knob_sweep*.json)
# round 4, GPU run 3: one rank of 8 at 2^24 -- reduction schedule variants on one resident key
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python scripts/dist_projection.py 24 8 7 points \
  "offmain:G16_REDUCE_OFF_MAIN=1,G16_BATCH_REDUCE=0;batch:G16_BATCH_REDUCE=1;default2:G16_PROJ_NOP=1;offmain2:G16_REDUCE_OFF_MAIN=1,G16_BATCH_REDUCE=0;batch2:G16_BATCH_REDUCE=1;default3:G16_PROJ_NOP=1" \
  > $O/proj_k24_sweep2.json 2> $O/proj_k24_sweep2.err; echo "rc=$?"; tail -3 $O/proj_k24_sweep2.err
python - $O/proj_k24_sweep2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("single", round(d["single_gpu_ms"], 2), d["single_msm"])
for k, v in d["ranks"].items():
    print(k, "rank ms", round(v["per_rank_ms"], 2), "eff", round(v["efficiency_before_xgmi"], 3), {a: b for a, b in v["stages_ms_alone"].items() if b})
PY
G16_PROJ_T1=137.0 timeout 900 python scripts/dist_projection.py 24 8 7 buckets "offmain:G16_REDUCE_OFF_MAIN=1" > $O/proj_k24_buckets.json 2> $O/proj_k24_buckets.err
python - $O/proj_k24_buckets.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["ranks"].items():
    print(k, "rank ms", round(v["per_rank_ms"], 2), v["ranks_timed"], {a: b for a, b in v["stages_ms_alone"].items() if b})
PY
