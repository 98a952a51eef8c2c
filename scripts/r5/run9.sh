#!/bin/bash
# round 5, GPU call 9: is the filtered B view a gain on the dense-skewed key (most points at infinity belong to 0 / 1 wires)?
O=gpurun_out/r5_9; mkdir -p $O; export TMPDIR=/tmp
for sb in 0 1 0 1; do
  G16_SPARSE_B=$sb timeout 600 python bench.py --workload dense-skewed --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/ds20_sb$sb.json 2> $O/ds20_sb$sb.err
  python -c "
import json; d=json.loads(open('$O/ds20_sb$sb.json').read().strip().splitlines()[-1]); print('dense-skewed 2^20 sparse_b=$sb', round(d['ms_per_step'],3), d['config']['msm'].get('sparse_b'), {k: round(v,2) for k,v in d['stages_ms_per_step'].items() if v})"
done
python - <<'PY'
import sys
sys.path.insert(0, '.')
import bench, circom_compat_amd as cc, random, numpy as np
mats, (A, B, Cm), w, n_vars = bench.dense_skewed_circuit(cc, 20)
inb = np.zeros(n_vars, dtype=bool); inb[B.col] = True
w = np.array([1 if x in (0, 1) else 0 for x in w])
print("wires", n_vars, "not in B", int((~inb).sum()), "of which 0/1-valued", int(((~inb) & (w == 1)).sum()))
PY
