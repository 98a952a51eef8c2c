#!/bin/bash
# round 3, GPU run 14: bucket reduction with 2 * run parked in LDS (G2: 478 registers, no spills, no scratch) vs the shipped kernel
cd /root/repo
export TMPDIR=/tmp
ab() { name=$1; shift
  env "$@" G16_BENCH_NO_PIPELINE=1 timeout 300 python bench.py --steps 16 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'ms/step', round(d['ms_per_step'],3), 'reduce', round(d['stages_ms_per_step']['msm_reduce'],3))"
}
ab base G16_X=0
ab lds G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_lds.so
ab base_b G16_X=0
ab lds_b G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_lds.so
cd /tmp; rm -rf /tmp/prof_r
G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_lds.so G16_BENCH_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_r -o st -- python /root/repo/bench.py --steps 8 --warmup 2 --cpu-log2 0 > /dev/null 2>&1
db=$(find /tmp/prof_r -name "*.db" | head -1)
python /root/repo/scripts/rocpd_stats.py $db | grep -E "k_bucket_reduce" | cut -c1-130
