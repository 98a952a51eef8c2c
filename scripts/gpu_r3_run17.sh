#!/bin/bash
# round 3, GPU run 17: one proof size above the largest configuration -- 2^25 constraints on ONE GPU
# ("full planes fit one GPU up to n = 2^25"); pairing-verified, no CPU leg
cd /root/repo
mkdir -p gpurun_out/run17
free -g | tee gpurun_out/run17/free.txt
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "$avail" -lt 96 ]; then echo "less than 96 GB of host memory available: not attempting 2^25"; exit 0; fi
G16_BENCH_NO_PIPELINE=1 timeout 900 python bench.py --log2 25 --steps 3 --warmup 1 --cpu-log2 0 \
  > gpurun_out/run17/r03_bench_chain25.json 2> gpurun_out/run17/err.txt
echo "rc=$?"; tail -5 gpurun_out/run17/err.txt
python - <<'PY'
import json
try:
    d = json.loads(open('/root/repo/gpurun_out/run17/r03_bench_chain25.json').read().strip().splitlines()[-1])
    print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "M/s", d.get("parity"), d.get("msm"), d.get("hbm_bytes"))
except Exception as e:
    print("no line", e)
PY
