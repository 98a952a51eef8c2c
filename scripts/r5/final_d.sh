#!/bin/bash
# round 5, end-of-round validation D: the full PMC counter set over the accumulation launches of the final binary
# (separate rocprofv3 --pmc passes, no trace domain) -> profiles/r05_pmc_k22_accumulate.txt, and the stamped
# profiles/pmc_traffic.json that bench.py falls back to when its own two passes cannot run
TAG=r05
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
sed -i 's/--cpu-log2 0 >/--cpu-log2 0 --no-pmc >/' scripts/pmc_passes.sh
bash scripts/pmc_passes.sh 22 final_$TAG/pmc > $O/pmc_passes.log 2>&1; tail -9 $O/pmc_passes.log | head -8
python scripts/pmc_traffic.py gpurun_out/final_$TAG/pmc 22 $O/pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/final_$TAG/pmc > $O/${TAG}_pmc_k22_accumulate.txt 2>&1; head -40 $O/${TAG}_pmc_k22_accumulate.txt
