#!/bin/bash
# round 6, end-of-round validation D: the full PMC counter set over the accumulation launches of the final binary ->
# profiles/r06_pmc_k22_accumulate.txt and the stamped profiles/pmc_traffic.json (bench.py's fallback when its own two
# passes cannot run AND the library hash matches)
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
rm -rf gpurun_out/final_$TAG/pmc
bash scripts/pmc_passes.sh 22 final_$TAG/pmc > $O/pmc_passes.log 2>&1; tail -9 $O/pmc_passes.log | head -8
python scripts/pmc_traffic.py gpurun_out/final_$TAG/pmc 22 $O/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final_$TAG/pmc > $O/${TAG}_pmc_k22_accumulate.txt 2>&1; head -12 $O/${TAG}_pmc_k22_accumulate.txt
