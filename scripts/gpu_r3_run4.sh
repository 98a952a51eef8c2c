#!/bin/bash
# round 3, GPU run 4: A/B of the rank schedule knobs on ONE box (2^22, 8 ranks, bucket mode), 2^24 projection + timeline
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python scripts/dist_projection.py 22 8 5 buckets > gpurun_out/r3_ab_$name.json 2> gpurun_out/r3_ab_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r3_ab_%s.json'%sys.argv[1]))
for k,v in d["ranks"].items():
    print(sys.argv[1], "T1", round(d["single_gpu_ms"],2), k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"])
PY
}
run default G16_X=0
T1=$(python -c "import json;print(json.load(open('/root/repo/gpurun_out/r3_ab_default.json'))['single_gpu_ms'])")
run nodefer G16_PROJ_T1=$T1 G16_MSM_AFTER_PHASE2=0
run grid4096 G16_PROJ_T1=$T1 G16_ACC_GRID=4096
run grid1024 G16_PROJ_T1=$T1 G16_ACC_GRID=1024
run redlanes G16_PROJ_T1=$T1 G16_RED_LANES=32768
timeout 1200 python scripts/dist_projection.py 24 8 3 buckets > gpurun_out/r3_proj_k24_c.json 2> gpurun_out/r3_proj_k24_c.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_proj_k24_c.json'))
print("k24 T1", d["single_gpu_ms"])
for k,v in d["ranks"].items():
    print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"], v["stages_ms_alone"])
PY
cd /tmp
for k in 22 24; do
rm -rf /tmp/prof_$k
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$k -o trace -- python /root/repo/scripts/dist_rank_trace.py $k 8 buckets 4 3 > /root/repo/gpurun_out/r3_trace3_$k.log 2>&1
db=$(find /tmp/prof_$k -name "*.db" | head -1)
python /root/repo/scripts/rocpd_timeline.py $db 150 > /root/repo/gpurun_out/r3_rank8_timeline3_k${k}_buckets.txt 2>&1
done
