#!/bin/bash
# round 3, GPU run 11: G1 kernels capped at 128 VGPRs (four waves per SIMD, 56 spilled VGPRs) vs the shipped 149-VGPR build, same box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { name=$1; shift
  env "$@" G16_BENCH_NO_PIPELINE=1 timeout 300 python bench.py --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab11_$name.json 2> gpurun_out/r3_ab11_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r3_ab11_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x["kernel"]:round(x["avg_launch_ms"],3) for x in d["roofline"]["all_accumulate_launches"]}
print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), k)
PY
}
ab base G16_X=0
ab w4_4096 G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_w4.so G16_ACC_GRID=4096
ab w4_3072 G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_w4.so
ab w4_2048 G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_w4.so G16_ACC_GRID=2048
ab base_b G16_X=0
