#!/bin/bash
# round 5, GPU call 14 (experiment, nothing shipped): the distributed witness map's three element-wise twiddle products
# as single-level per-rank tables, two of them fused into the closing product of the local DIF in front of them
# (k_dist_twist gone, k_dist_pack1 a pure transpose): one rank of 8 at 2^24 alone on one GPU, product vs variant
O=gpurun_out/r5_14; mkdir -p $O; export TMPDIR=/tmp
for v in product variant; do
  if [ $v = variant ]; then export G16_AMD_LIB=$PWD/scripts/variants/libg16_distfuse.so; else unset G16_AMD_LIB; fi
  timeout 900 python scripts/dist_projection.py 24 8 7 points > $O/proj_k24_$v.json 2> $O/proj_k24_$v.err
  python -c "
import json; d=json.loads(open('$O/proj_k24_$v.json').read().strip().splitlines()[-1]); print('$v single', round(d['single_gpu_ms'],2), {k:(round(x['per_rank_ms'],3), round(x['efficiency_before_xgmi'],4)) for k,x in d['ranks'].items()})"
done
