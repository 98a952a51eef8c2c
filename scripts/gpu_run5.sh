cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=" | head -30
python scripts/dist_projection.py 22 8 7 > gpurun_out/r02_rank8_ab0.json 2>/dev/null; cat gpurun_out/r02_rank8_ab0.json; echo
T1=$(python -c "import json;print(json.load(open('gpurun_out/r02_rank8_ab0.json'))['single_gpu_ms'])")
for rep in 1 2 3; do for prio in 0 1; do
  echo -n "xs_prio=$prio rep=$rep: "; G16_PROJ_XS_PRIO=$prio G16_PROJ_T1=$T1 python scripts/dist_projection.py 22 8 7 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ranks']['8']['per_rank_ms'])"
done; done
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^=" | head -20
