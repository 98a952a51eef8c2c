cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/dist_projection.py 22 2,4,8 7 > gpurun_out/r02_proj_k22_prio.json 2>/dev/null; cat gpurun_out/r02_proj_k22_prio.json; echo
T1=$(python -c "import json;print(json.load(open('gpurun_out/r02_proj_k22_prio.json'))['single_gpu_ms'])")
cd /tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/rank_tl2 -o rank8 -- env G16_PROJ_T1=$T1 python $GRAFT_REPO_ROOT/scripts/dist_projection.py 22 8 3 > $GRAFT_REPO_ROOT/gpurun_out/rank_tl2.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/rank_tl2 -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB 130 > gpurun_out/r02_rank8_timeline_prio.txt
G16_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 8 --log2 20 --steps 3 --cpu-log2 14 > gpurun_out/r02_bench_inlib8_onegpu.json 2> gpurun_out/r02_bench_inlib8_onegpu.err; echo "inlib8 rc=$?"; head -c 600 gpurun_out/r02_bench_inlib8_onegpu.json
