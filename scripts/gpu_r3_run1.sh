#!/bin/bash
# round 3, GPU run 1: bucket-range sharding -- GPU suite + strong-scaling projections (same box)
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_run1_pytest.log 2>&1
tail -5 gpurun_out/r3_run1_pytest.log
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > gpurun_out/r3_proj_k22.json 2> gpurun_out/r3_proj_k22.err
tail -c 3000 gpurun_out/r3_proj_k22.json
timeout 1200 python scripts/dist_projection.py 24 8 3 points,buckets > gpurun_out/r3_proj_k24.json 2> gpurun_out/r3_proj_k24.err
tail -c 2000 gpurun_out/r3_proj_k24.json
tail -3 gpurun_out/r3_proj_k24.err
