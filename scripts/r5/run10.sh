#!/bin/bash
# round 5, GPU call 10: filtered view for the G2 query only (B1 stays in the A|B1 pair launch): parity + A/B on both sparse-B workloads
O=gpurun_out/r5_10; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "sparse_b or real_poseidon or dense_skewed_2p20 or (headline_sizes and not 24)" > $O/pytest_sparse.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sparse.log
tail -3 $O/pytest_sparse.log
for wl in poseidon dense-skewed; do
for sb in 0 1 0 1; do
  G16_SPARSE_B=$sb timeout 600 python bench.py --workload $wl --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/${wl}_sb$sb.json 2> $O/${wl}_sb$sb.err
  python -c "
import json; d=json.loads(open('$O/${wl}_sb$sb.json').read().strip().splitlines()[-1]); print('$wl 2^20 sparse_b=$sb', round(d['ms_per_step'],3), d['config']['msm'].get('sparse_b'), {k: round(v,2) for k,v in d['stages_ms_per_step'].items() if v})"
done
done
