#!/bin/bash
# round 5, end-of-round validation F: the whole GPU suite once more at HEAD (after the 240-row Poseidon generator and
# the forced-tables test at 2^15 went in; library unchanged: 098abcb74168b5b8)
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/${TAG}_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -14 $O/${TAG}_pytest_gpu_final.log
