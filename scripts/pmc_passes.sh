#!/bin/bash
# PMC passes over the MSM bucket kernel (separate runs per counter group, no trace domains mixed in:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: scripts/pmc_passes.sh <log2> <out-subdir> [kernel-regex]
set -u
K=${1:-22}; OUT=${2:-pmc}; RE=${3:-k_bucket_accumulate}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_IFETCH GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  i=$((i+1))
  G16_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "$RE" -f csv -d $ROOT/gpurun_out/$OUT/p$i -o p$i -- \
      python $ROOT/bench.py --log2 $K --steps 1 --warmup 0 --cpu-log2 0 --no-pmc --no-secondary > $ROOT/gpurun_out/$OUT/p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
find $ROOT/gpurun_out/$OUT -name "*.csv" | head -20
