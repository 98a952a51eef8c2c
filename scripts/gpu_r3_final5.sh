#!/bin/bash
# round 3: the definitive binary once more on another box (the final4 box ran everything 15-45 % slower:
# DESIGN.md section 6 "fresh boxes differ"); bench line, kernel trace, projections, rank timelines
cd /root/repo
mkdir -p gpurun_out/final5
OUT=/root/repo/gpurun_out/final5
export TMPDIR=/tmp
sha256sum circom_compat_amd/libg16_amd.so | cut -c1-16 > $OUT/library_sha16.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/final5/r03_bench_default.json').read().strip().splitlines()[-1])
print("bench", round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,2), "M/s; pcie", d["ms_per_step_pcie_inclusive"], "cpu", d["cpu_baseline"]["value"], "pipelined", d.get("value_pipelined",{}).get("ms_per_proof"))
for x in d["roofline"]["all_accumulate_launches"]: print(x["kernel"], round(x["avg_launch_ms"],3), x["frac"], x["traffic"])
PY
cd /tmp; rm -rf /tmp/prof_f
G16_BENCH_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o st -- python /root/repo/bench.py --steps 12 --warmup 3 --cpu-log2 0 > $OUT/kernel_trace.log 2>&1
db=$(find /tmp/prof_f -name "*.db" | head -1)
python /root/repo/scripts/rocpd_stats.py $db > $OUT/r03_k22_kernel_stats.txt 2>&1
grep -E "k_bucket_accumulate|k_acc_fixup" $OUT/r03_k22_kernel_stats.txt | cut -c1-150
cd /root/repo
timeout 600 python bench.py --log2 20 --steps 10 --warmup 2 > $OUT/r03_bench_chain20.json 2> /dev/null
python -c "import json; d=json.loads(open('$OUT/r03_bench_chain20.json').read().strip().splitlines()[-1]); print('k20', round(d['ms_per_step'],3))"
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > $OUT/r03_proj_k22.json 2> $OUT/r03_proj_k22.err
timeout 1500 python scripts/dist_projection.py 24 8 3 points,buckets > $OUT/r03_proj_k24.json 2> $OUT/r03_proj_k24.err
python - <<'PY'
import json
for f in ("r03_proj_k22","r03_proj_k24"):
    d=json.load(open('/root/repo/gpurun_out/final5/%s.json'%f))
    print(f, round(d["single_gpu_ms"],2))
    for k,v in d["ranks"].items():
        print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), "w/link", round(v["efficiency_if_all_link_time_exposed"],3), v["ranks_timed"])
PY
cd /tmp
for cfg in "22 points 0" "22 buckets 4" "24 points 0"; do
  set -- $cfg
  rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_t -o trace -- python /root/repo/scripts/dist_rank_trace.py $1 8 $2 $3 3 > $OUT/trace_$1_$2.log 2>&1
  db=$(find /tmp/prof_t -name "*.db" | head -1)
  python /root/repo/scripts/rocpd_timeline.py $db 150 > $OUT/r03_rank8_timeline_k$1_$2.txt 2>&1
done
