#!/bin/bash
# round 3, GPU run 5: the new level-1 sort -- GPU tests that sort, bench, projections
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_kernels.py -m gpu -x -q -k "msm or shard or multi or determinism or dense or config5" > gpurun_out/r3_run5_pytest.log 2>&1
tail -3 gpurun_out/r3_run5_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-log2 0 > gpurun_out/r3_run5_bench.json 2> gpurun_out/r3_run5_bench.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r3_run5_bench.json').read().strip().splitlines()[-1])
print("bench ms/step", d["ms_per_step"], d["stages_ms_per_step"])
print(json.dumps(d["roofline"])[:1500])
PY
run() { name=$1; shift
  env "$@" timeout 600 python scripts/dist_projection.py 22 8 5 buckets > gpurun_out/r3_ab5_$name.json 2> gpurun_out/r3_ab5_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r3_ab5_%s.json'%sys.argv[1]))
for k,v in d["ranks"].items():
    print(sys.argv[1], "T1", round(d["single_gpu_ms"],2), k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"], v["stages_ms_alone"])
PY
}
run default G16_X=0
T1=$(python -c "import json;print(json.load(open('/root/repo/gpurun_out/r3_ab5_default.json'))['single_gpu_ms'])")
run defer G16_PROJ_T1=$T1 G16_MSM_AFTER_PHASE2=1
timeout 1200 python scripts/dist_projection.py 24 8 3 buckets > gpurun_out/r3_proj_k24_d.json 2> gpurun_out/r3_proj_k24_d.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_proj_k24_d.json'))
print("k24 T1", d["single_gpu_ms"])
for k,v in d["ranks"].items():
    print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"], v["stages_ms_alone"])
PY
cd /tmp
rm -rf /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s -o st -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-log2 0 > /root/repo/gpurun_out/r3_run5_stats.log 2>&1
db=$(find /tmp/prof_s -name "*.db" | head -1)
python /root/repo/scripts/rocpd_stats.py $db > /root/repo/gpurun_out/r3_run5_kernel_stats.txt 2>&1
grep -E "k_part|k_bucket_(count|scatter)|k_scan|k_pick" /root/repo/gpurun_out/r3_run5_kernel_stats.txt | cut -c1-150
