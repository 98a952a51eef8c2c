#!/bin/bash
# round 3, GPU run 6: where does the level-1 scatter's time go?  A/B of partition count / grid on one box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
stats() { name=$1; shift
  cd /tmp; rm -rf /tmp/prof_$name
  env "$@" timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$name -o st -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-log2 0 > /root/repo/gpurun_out/r3_run6_$name.log 2>&1
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  python /root/repo/scripts/rocpd_stats.py $db > /root/repo/gpurun_out/r3_run6_stats_$name.txt 2>&1
  echo "== $name"; grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r3_run6_$name.log | head -1
  grep -E "k_part|k_bucket_(count|scatter)" /root/repo/gpurun_out/r3_run6_stats_$name.txt | cut -c1-120
  cd /root/repo
}
stats base G16_X=0
stats bins8 G16_SORT_BINS=8
stats bins8_grid512 G16_SORT_BINS=8 G16_SORT_GRID=512
stats grid512 G16_SORT_GRID=512
stats bins9_grid1024 G16_SORT_BINS=9 G16_SORT_GRID=1024
# PMC on the level-1 scatter (separate passes)
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT TCC_MISS TCC_REQ" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_128B"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "k_part_scatter|k_bucket_scatter" -f csv -d /root/repo/gpurun_out/r3_pmc_sort/p$i -o p$i -- \
      python /root/repo/bench.py --steps 1 --warmup 0 --cpu-log2 0 > /root/repo/gpurun_out/r3_pmc_sort_p$i.log 2>&1
done
python /root/repo/scripts/pmc_summary.py /root/repo/gpurun_out/r3_pmc_sort 2>&1 | head -70
