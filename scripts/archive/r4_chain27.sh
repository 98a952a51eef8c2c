#!/bin/bash
# round 4: 2^27 constraints on ONE GPU -- the largest domain the reference accepts (Fr two-adicity 28: the
# 2n-domain of qap.rs:63-68 is the whole two-adic subgroup).  Key from the GPU trapdoor generator (its size-2^28
# transform), planes per the memory plan, pairing check, bytes against the CPU restatement (--cpu-own).
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4_chain27; mkdir -p $O; export TMPDIR=/tmp
free -g | head -2
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "$avail" -lt 400 ]; then echo "less than 400 GB of host memory available: not attempting 2^27"; exit 0; fi
G16_BENCH_NO_PIPELINE=1 timeout 2300 python bench.py --log2 27 --steps 2 --warmup 1 ${1:---cpu-own} > $O/r04_bench_chain27.json 2> $O/chain27.err
echo "rc=$?"; tail -8 $O/chain27.err | cut -c1-300
python - $O/r04_bench_chain27.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "M/s", d.get("parity"), d["config"].get("msm"), d["stages_ms_per_step"], (d.get("cpu_baseline") or {}).get("samples_s"), "setup_s", d.get("setup_s"))
except Exception as e:
    print("no line", e)
PY
