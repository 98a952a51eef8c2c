#!/bin/bash
# round 3, last pass on the committed binary: parity suite, PMC passes (traffic stamped with this library), the
# headline bench line, a kernel trace without overlapping proofs
cd /root/repo
mkdir -p gpurun_out/final2
OUT=/root/repo/gpurun_out/final2
export TMPDIR=/tmp
sha256sum circom_compat_amd/libg16_amd.so | cut -c1-16 > $OUT/library_sha16.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash scripts/pmc_passes.sh 22 final2/pmc22 k_bucket_accumulate > $OUT/pmc_passes.log 2>&1
python scripts/pmc_traffic.py gpurun_out/final2/pmc22 22 profiles/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final2/pmc22 > $OUT/r03_pmc_k22_accumulate.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/final2/r03_bench_default.json').read().strip().splitlines()[-1])
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
