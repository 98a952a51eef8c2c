#!/bin/bash
# round 3, GPU run 9: G1 at 3072 workgroups / G2 at 2048 -- A/B + MSM parity tests
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_kernels.py -m gpu -x -q -k "msm or determinism or dense or config5 or multi or shard" > gpurun_out/r3_run9_pytest.log 2>&1
tail -3 gpurun_out/r3_run9_pytest.log
ab() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab9_$name.json 2> gpurun_out/r3_ab9_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r3_ab9_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x["kernel"]:round(x["avg_launch_ms"],3) for x in d["roofline"]["all_accumulate_launches"]}
print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), k, "reduce", round(d["stages_ms_per_step"]["msm_reduce"],2), "sort", round(d["stages_ms_per_step"]["msm_sort"],2))
PY
}
ab new G16_X=0
ab exact2048 G16_ACC_FAST=0 G16_ACC_GRID=2048
ab fast2048 G16_ACC_GRID=2048
ab new_b G16_X=0
ab g1_4608 G16_ACC_GRID=4608
ab k20 G16_X=0
timeout 300 python bench.py --log2 20 --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab9_k20.json 2>/dev/null
G16_ACC_FAST=0 G16_ACC_GRID=2048 timeout 300 python bench.py --log2 20 --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab9_k20_old.json 2>/dev/null
python - <<'PY'
import json
for n in ("k20","k20_old"):
    d=json.loads(open('/root/repo/gpurun_out/r3_ab9_%s.json'%n).read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],3))
PY
