#!/bin/bash
# round 6, validation I: the GPU suite (driver's command) + smoke at HEAD after the checker's thread pool change
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_pytest_gpu_final.log 2>&1 ) 2>&1 | tail -3; tail -18 $O/${TAG}_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
