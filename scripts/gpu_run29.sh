# rank of 8 at 2^24 (2^21 points, 2^19 buckets): where do the reductions go? (same box, T1 measured once)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/dist_projection.py 24 8 5 > gpurun_out/r29_base.json 2>/dev/null; cat gpurun_out/r29_base.json; echo
T1=$(python -c "import json; print(json.load(open('gpurun_out/r29_base.json'))['single_gpu_ms'])")
for knob in "G16_B2_RED_STREAM=1" "G16_BATCH_REDUCE=0 G16_REDUCE_OFF_MAIN=1" "G16_BATCH_REDUCE=0 G16_REDUCE_OFF_MAIN=1 G16_B2_RED_STREAM=1" "G16_B2_RED_STREAM=0"; do
  echo -n "$knob: "; env $knob G16_PROJ_T1=$T1 python scripts/dist_projection.py 24 8 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ranks']['8']['per_rank_ms'], d['ranks']['8']['efficiency_before_xgmi'])"
done
