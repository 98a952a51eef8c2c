#!/bin/bash
# round 3, GPU run 13: same-box A/B -- fix-up inline vs on the reducing stream, hidden reductions at 65536 / 16384 / 8192 threads
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
one() { name=$1; shift
  a=$(env "$@" G16_BENCH_NO_PIPELINE=1 timeout 300 python bench.py --log2 20 --steps 16 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))")
  b=$(env "$@" G16_PROJ_T1=37.0 timeout 600 python scripts/dist_projection.py 22 8 7 points 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ranks']['points:8']['per_rank_ms'],3))")
  c=$(env "$@" G16_PROJ_T1=37.0 timeout 600 python scripts/dist_projection.py 22 8 7 buckets 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ranks']['buckets:8']['ranks_timed'])")
  echo "$name  k20 $a  points:8 rank $b  buckets:8 ranks $c"
}
one inline_65536 G16_FIXUP_INLINE=1 G16_RED_LANES_HIDDEN=65536
one red_65536 G16_RED_LANES_HIDDEN=65536
one red_8192 G16_X=0
one inline_8192 G16_FIXUP_INLINE=1
one red_16384 G16_RED_LANES_HIDDEN=16384
one inline_16384 G16_FIXUP_INLINE=1 G16_RED_LANES_HIDDEN=16384
one inline_65536_b G16_FIXUP_INLINE=1 G16_RED_LANES_HIDDEN=65536
one red_8192_b G16_X=0
