# projected strong scaling with the round's final binary (one rank alone on the GPU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/dist_projection.py 22 2,4,8 7 > gpurun_out/r02_proj_k22_final.json 2> gpurun_out/r02_proj_k22_final.err; cat gpurun_out/r02_proj_k22_final.json; echo
timeout 900 python scripts/dist_projection.py 24 8 3 > gpurun_out/r02_proj_k24_final.json 2> gpurun_out/r02_proj_k24_final.err; cat gpurun_out/r02_proj_k24_final.json; tail -2 gpurun_out/r02_proj_k24_final.err
