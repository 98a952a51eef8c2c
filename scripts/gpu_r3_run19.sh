#!/bin/bash
# round 3, GPU run 19: 2^26 constraints on ONE GPU -- past the full-plane capacity point, the maximum
# size of the 26-bit point index (fit_config falls back to planes < W, bucket sets folded by k_horner)
cd /root/repo
mkdir -p gpurun_out/run19
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "$avail" -lt 200 ]; then echo "less than 200 GB of host memory available: not attempting 2^26"; exit 0; fi
G16_BENCH_NO_PIPELINE=1 timeout 1000 python bench.py --log2 26 --steps 2 --warmup 1 --cpu-log2 0 \
  > gpurun_out/run19/r03_bench_chain26.json 2> gpurun_out/run19/err.txt
echo "rc=$?"; tail -5 gpurun_out/run19/err.txt
python - <<'PY'
import json
try:
    d = json.loads(open('/root/repo/gpurun_out/run19/r03_bench_chain26.json').read().strip().splitlines()[-1])
    print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "M/s", d.get("parity"), d["config"]["msm"], d["stages_ms_per_step"])
except Exception as e:
    print("no line", e)
PY
