#!/bin/bash
# round 3, GPU run 15: level-1 partition pass staged through LDS (k_part_scatter_staged) vs the direct
# scatter (G16_SORT_STAGED=0), same binary: proof time, sort stage, kernel durations, parity tests
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/run15
timeout 400 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -3
ab() { name=$1; shift
  env "$@" G16_BENCH_NO_PIPELINE=1 timeout 300 python bench.py --steps 16 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stages_ms_per_step']
print('$name', 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in st.items() if 'sort' in k})"
}
ab direct G16_SORT_STAGED=0
ab staged G16_SORT_STAGED=1
ab direct_b G16_SORT_STAGED=0
ab staged_b G16_SORT_STAGED=1
for v in 0 1; do
  cd /tmp; rm -rf /tmp/prof_s
  G16_SORT_STAGED=$v G16_BENCH_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s -o st -- python /root/repo/bench.py --steps 8 --warmup 2 --cpu-log2 0 > /dev/null 2>&1
  db=$(find /tmp/prof_s -name "*.db" | head -1)
  echo "staged=$v"
  python /root/repo/scripts/rocpd_stats.py $db | grep -E "k_part_|k_bucket_count|k_bucket_scatter" | cut -c1-140 | tee /root/repo/gpurun_out/run15/sort_kernels_staged$v.txt
done
cd /root/repo
G16_SORT_STAGED=1 timeout 300 python bench.py --log2 20 --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k20 staged', round(d['ms_per_step'],3))"
G16_SORT_STAGED=0 timeout 300 python bench.py --log2 20 --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k20 direct', round(d['ms_per_step'],3))"
