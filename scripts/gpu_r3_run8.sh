#!/bin/bash
# round 3, GPU run 8: optimistic G1 accumulation (deferred exact additions) -- parity suite + same-box A/B
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_run8_pytest.log 2>&1
tail -4 gpurun_out/r3_run8_pytest.log
ab() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --cpu-log2 0 > gpurun_out/r3_ab8_$name.json 2> gpurun_out/r3_ab8_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r3_ab8_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x["kernel"]:round(x["avg_launch_ms"],3) for x in d["roofline"]["all_accumulate_launches"]}
print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), k, "reduce", round(d["stages_ms_per_step"]["msm_reduce"],2))
PY
}
ab exact G16_ACC_FAST=0
ab fast2048 G16_ACC_FAST=1
ab fast1536 G16_ACC_FAST=1 G16_ACC_GRID=1536
ab fast3072 G16_ACC_FAST=1 G16_ACC_GRID=3072
ab exact_b G16_ACC_FAST=0
ab fast2048_b G16_ACC_FAST=1
