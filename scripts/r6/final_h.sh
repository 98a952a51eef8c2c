#!/bin/bash
# round 6, validation H: the driver's bench command after the CPU restatement's thread pool was sized to the host's
# cgroup CPU grant (16 of the 256 visible CPUs on a gpurun box: 32 threads); same library binary
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default_cpu_grant.json 2> $O/bench_default_h.err ) 2>&1 | tail -3
python - $O/${TAG}_bench_default_cpu_grant.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["cpu_baseline"]
print(d["ms_per_step"], d["value"], d.get("clock_mhz"), d["parity"])
print({k: c[k] for k in c if k != "sample"})
print(d.get("gpu_over_cpu"), d.get("gpu_over_cpu_all_cores"))
PY
