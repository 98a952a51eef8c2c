# per-rank behaviour of the sharded prover (rank 0 of 8 alone on the GPU): timeline + scheduling knobs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/dist_projection.py 22 8 7 > gpurun_out/r02_rank8_base.json 2>/dev/null; cat gpurun_out/r02_rank8_base.json; echo
T1=$(python -c "import json;print(json.load(open('gpurun_out/r02_rank8_base.json'))['single_gpu_ms'])")
for knob in "G16_ACC_GRID=1024" "G16_ACC_GRID=4096" "G16_ACC_GRID=8192" "G16_MSM_CU_RESERVE=16" "G16_MSM_CU_RESERVE=32" "G16_MSM_CU_RESERVE=16 G16_ACC_GRID=4096" "G16_AUX_PRIORITY=0"; do
  echo "== $knob"; env $knob G16_PROJ_T1=$T1 python scripts/dist_projection.py 22 8 7 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ranks'])"
done
cd /tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/rank_tl -o rank8 -- env G16_PROJ_T1=$T1 python $GRAFT_REPO_ROOT/scripts/dist_projection.py 22 8 3 > $GRAFT_REPO_ROOT/gpurun_out/rank_tl.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/rank_tl -name "*.db" | head -1); echo "db=$DB"
python scripts/rocpd_timeline.py $DB 130 > gpurun_out/r02_rank8_timeline.txt; head -5 gpurun_out/r02_rank8_timeline.txt
