#!/bin/bash
# round 5, GPU call 11 (experiment, nothing shipped): single-level NTT tables up to 2^27 (a variant build,
# scripts/variants/libg16_ntt_tables27.so = the product sources with NTT29_FULL_TABLE_MAX_LOG = 27) against the
# product library at 2^26, same box: does the witness map -- what the H MSM waits for at this size -- get
# shorter, and does the 6.4 GB of tables cost a plane?
O=gpurun_out/r5_11; mkdir -p $O; export TMPDIR=/tmp
for v in product variant; do
  if [ $v = variant ]; then export G16_AMD_LIB=$PWD/scripts/variants/libg16_ntt_tables27.so; else unset G16_AMD_LIB; fi
  timeout 900 python bench.py --log2 26 --steps 3 --warmup 1 --cpu-log2 0 --no-pmc > $O/k26_$v.json 2> $O/k26_$v.err
  python -c "
import json; d=json.loads(open('$O/k26_$v.json').read().strip().splitlines()[-1]); m=d['config']['msm']; print('2^26 $v', round(d['ms_per_step'],2), 'ms', d['parity'], 'planes_w', m['planes_w'], 'D_w', m['D_w'], 'planes_h', m['planes_h'], {k: round(x,1) for k,x in d['stages_ms_per_step'].items() if x})"
done
