# same-box A/B of the field29 variants + strong-scaling projections (one rank alone on the GPU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for v in o0c0 o1c0 o2c0 o0c1 o2c1; do
    G16_AMD_LIB=$PWD/circom_compat_amd/libg16_var_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --cpu-log2 0 > gpurun_out/r02_ab_${v}_$rep.json 2> gpurun_out/r02_ab_${v}_$rep.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r02_ab_${v}_$rep.json"))
s=d["stages_ms_per_step"]
print("$v rep$rep ms=%.3f acc_g1=%.3f acc_g2=%.3f wm=%.2f red=%.2f sort=%.2f" % (d["ms_per_step"], s["msm_accumulate_g1"], s["msm_accumulate_g2"], s["witness_map"], s["msm_reduce"], s["msm_sort"]))
PY
  done
done
for v in o0c0 o2c1; do
  G16_AMD_LIB=$PWD/circom_compat_amd/libg16_var_$v.so python scripts/alu_bench.py > gpurun_out/r02_alu_$v.txt 2>&1; tail -2 gpurun_out/r02_alu_$v.txt
done
timeout 600 python scripts/dist_projection.py 22 2,4,8 5 > gpurun_out/r02_proj_k22.json 2> gpurun_out/r02_proj_k22.err; cat gpurun_out/r02_proj_k22.json
timeout 900 python scripts/dist_projection.py 24 8 3 > gpurun_out/r02_proj_k24.json 2> gpurun_out/r02_proj_k24.err; cat gpurun_out/r02_proj_k24.json; tail -3 gpurun_out/r02_proj_k24.err
