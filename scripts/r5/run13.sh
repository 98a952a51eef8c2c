#!/bin/bash
# round 5, GPU call 13 (experiment, nothing shipped): the G2 accumulation with the accumulator's zz / zzz parked in LDS
# (VERDICT r4 item 6), on variant builds: (a) optimistic G2 kernel + parking, registers as the compiler likes
# (256 V + 81 A, no scratch, one wave per SIMD); (b) the same capped at 256 registers = two waves per SIMD (99 VGPRs
# spilled, 332 B of scratch).  The B2 launch (stage msm_accumulate_g2) against the shipped exact kernel, same box.
O=gpurun_out/r5_13; mkdir -p $O; export TMPDIR=/tmp
for v in product park park_w2 product park park_w2; do
  case $v in product) unset G16_AMD_LIB;; park) export G16_AMD_LIB=$PWD/scripts/variants/libg16_g2park.so;; park_w2) export G16_AMD_LIB=$PWD/scripts/variants/libg16_g2park_w2.so;; esac
  timeout 600 python bench.py --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $O/b22_$v.json 2> $O/b22_$v.err
  python -c "
import json; d=json.loads(open('$O/b22_$v.json').read().strip().splitlines()[-1]); st=d['stages_ms_per_step']; print('2^22 $v', round(d['ms_per_step'],2), 'ms; B2 launch', round(st['msm_accumulate_g2'],2), 'pair', round(st['msm_accumulate_g1_pair'],2), 'L+H', round(st['msm_accumulate_g1'],2), 'fixup', round(st['msm_fixup'],2), d['parity'])"
done
