#!/bin/bash
# round 3, GPU run 10: optimistic kernel for the G2 launch (279 VGPRs, 1 wave/SIMD) and capped at 256 VGPRs (2 waves/SIMD); pipelined mode; 2^20 check
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
ab() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab10_$name.json 2> gpurun_out/r3_ab10_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r3_ab10_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x["kernel"]:round(x["avg_launch_ms"],3) for x in d["roofline"]["all_accumulate_launches"]}
print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), k, "pipelined", d.get("value_pipelined"))
PY
}
ab base G16_X=0
ab g2fast G16_ACC_FAST_G2=1
ab g2fast_w2 G16_ACC_FAST_G2=1 G16_AMD_LIB=/root/repo/circom_compat_amd/libg16_w2.so
ab base_b G16_X=0
ab g2fast_b G16_ACC_FAST_G2=1
timeout 300 python bench.py --log2 20 --steps 12 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k20', round(d['ms_per_step'],3), d.get('value_pipelined'))"
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "sibling" 2>&1 | tail -2
