#!/bin/bash
# round 5, GPU call 12: where the fixed-base tables stop paying -- chain circuits 2^12 .. 2^15 with the tables forced
# on / off on the shipped binary (2^15 is one size above the automatic rule: 51 GB of tables), same box
O=gpurun_out/r5_12; mkdir -p $O; export TMPDIR=/tmp
for k in 12 13 14 15; do
for t in off on; do
  timeout 600 python bench.py --log2 $k --tables $t --steps 100 --warmup 10 --no-pmc --cpu-log2 0 > $O/chain${k}_$t.json 2> $O/chain${k}_$t.err
  python -c "
import json; d=json.loads(open('$O/chain${k}_$t.json').read().strip().splitlines()[-1]); print('chain 2^$k tables=$t', round(d['ms_per_step'],3), 'ms; host witness', round(d['ms_per_step_pcie_inclusive'],3), 'setup_s', round(d['setup_s'],2), d['config']['msm']['fixed_tables'], d['parity'])" 2>&1 | tail -1
done
done
