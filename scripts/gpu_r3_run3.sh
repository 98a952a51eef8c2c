#!/bin/bash
# round 3, GPU run 3: sharded-path tests, projections after (H by points, streamed digits), quick bench
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_kernels.py -m gpu -x -q -k "shard or multi or rank or rccl or bucket_range or msm or config5" > gpurun_out/r3_run3_pytest.log 2>&1
tail -5 gpurun_out/r3_run3_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-log2 0 > gpurun_out/r3_run3_bench.json 2> gpurun_out/r3_run3_bench.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r3_run3_bench.json').read().strip().splitlines()[-1])
print("bench ms/step", d["ms_per_step"], d["stages_ms_per_step"])
PY
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 buckets > gpurun_out/r3_proj_k22_b.json 2> gpurun_out/r3_proj_k22_b.err
timeout 1200 python scripts/dist_projection.py 24 8 3 buckets > gpurun_out/r3_proj_k24_b.json 2> gpurun_out/r3_proj_k24_b.err
python - <<'PY'
import json
for f in ("r3_proj_k22_b","r3_proj_k24_b"):
    d=json.load(open('/root/repo/gpurun_out/%s.json'%f))
    print(f, d["single_gpu_ms"])
    for k,v in d["ranks"].items():
        print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"], v["stages_ms_alone"])
PY
cd /tmp
rm -rf /tmp/prof_22
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_22 -o trace -- python /root/repo/scripts/dist_rank_trace.py 22 8 buckets 4 3 > /root/repo/gpurun_out/r3_trace2_22.log 2>&1
db=$(find /tmp/prof_22 -name "*.db" | head -1)
python /root/repo/scripts/rocpd_timeline.py $db 150 > /root/repo/gpurun_out/r3_rank8_timeline2_k22_buckets.txt 2>&1
