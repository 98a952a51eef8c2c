#!/bin/bash
# round 3, definitive pass on the committed binary: parity suite, PMC passes (traffic stamped with this library), the
# headline bench line, a kernel trace without overlapping proofs
cd /root/repo
mkdir -p gpurun_out/final4
OUT=/root/repo/gpurun_out/final4
export TMPDIR=/tmp
sha256sum circom_compat_amd/libg16_amd.so | cut -c1-16 > $OUT/library_sha16.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash scripts/pmc_passes.sh 22 final4/pmc22 k_bucket_accumulate > $OUT/pmc_passes.log 2>&1
python scripts/pmc_traffic.py gpurun_out/final4/pmc22 22 profiles/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final4/pmc22 > $OUT/r03_pmc_k22_accumulate.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/final4/r03_bench_default.json').read().strip().splitlines()[-1])
print("bench", round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,2), "M/s; pcie", d["ms_per_step_pcie_inclusive"], "cpu", d["cpu_baseline"]["value"], "pipelined", d.get("value_pipelined",{}).get("ms_per_proof"))
for x in d["roofline"]["all_accumulate_launches"]: print(x["kernel"], round(x["avg_launch_ms"],3), x["frac"], x["traffic"])
print(d["roofline"]["traffic_source"])
PY
cd /tmp; rm -rf /tmp/prof_f
G16_BENCH_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o st -- python /root/repo/bench.py --steps 12 --warmup 3 --cpu-log2 0 > $OUT/kernel_trace.log 2>&1
db=$(find /tmp/prof_f -name "*.db" | head -1)
python /root/repo/scripts/rocpd_stats.py $db > $OUT/r03_k22_kernel_stats.txt 2>&1
grep -E "k_bucket_accumulate|k_acc_fixup" $OUT/r03_k22_kernel_stats.txt | cut -c1-150
cd /root/repo
timeout 600 python bench.py --log2 20 --steps 10 --warmup 2 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k20', round(d['ms_per_step'],3))"
timeout 900 python scripts/dist_projection.py 22 2,4,8 5 points,buckets > $OUT/r03_proj_k22.json 2> $OUT/r03_proj_k22.err
timeout 1500 python scripts/dist_projection.py 24 8 3 points,buckets > $OUT/r03_proj_k24.json 2> $OUT/r03_proj_k24.err
python - <<'PY'
import json
for f in ("r03_proj_k22","r03_proj_k24"):
    d=json.load(open('/root/repo/gpurun_out/final4/%s.json'%f))
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
cd /root/repo
timeout 600 python bench.py --log2 20 --steps 10 --warmup 2 > $OUT/r03_bench_chain20.json 2> /dev/null
timeout 600 python bench.py --log2 20 --mode parts --steps 5 --warmup 1 --cpu-log2 0 > $OUT/r03_bench_parts_k20.json 2> /dev/null
timeout 600 python bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 > $OUT/r03_bench_poseidon20.json 2> /dev/null
timeout 600 python bench.py --workload dense-skewed --log2 20 --steps 10 --warmup 2 > $OUT/r03_bench_dense_skewed20.json 2> /dev/null
python - <<'PY'
import json
for n in ("chain20","parts_k20","poseidon20","dense_skewed20"):
    d=json.loads(open('/root/repo/gpurun_out/final4/r03_bench_%s.json'%n).read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],3), "ms pcie", d["ms_per_step_pcie_inclusive"], d["parity"])
PY
