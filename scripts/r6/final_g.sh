#!/bin/bash
# round 6, end-of-round validation G: the 2^26 proof of the FINAL binary byte-compared with the CPU restatement
# (opt-in test: ~4 min of CPU proof on the box's host cores)
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
( time G16_TEST_2P26=1 G16_TEST_2P26_BYTES=1 timeout 1500 python -m pytest tests/test_gpu_large.py -m gpu -q -s -k test_domain_2p26 > $O/${TAG}_pytest_2p26_bytes.log 2>&1 ) 2>&1 | tail -3
tail -5 $O/${TAG}_pytest_2p26_bytes.log | cut -c1-400
