#!/bin/bash
# round 5, GPU call 6: filtered B view (sparse B queries) -- parity at the three schedule branches + the Poseidon line A/B
O=gpurun_out/r5_6; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "sparse_b or real_poseidon or (headline_sizes and not 24)" --durations=8 > $O/pytest_sparse.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sparse.log
tail -12 $O/pytest_sparse.log
for sb in 0 1 0 1; do
  G16_SPARSE_B=$sb timeout 600 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 --no-pmc --cpu-log2 0 > $O/pos20_sb$sb.json 2> $O/pos20_sb$sb.err
  python -c "
import json; d=json.loads(open('$O/pos20_sb$sb.json').read().strip().splitlines()[-1]); print('poseidon 2^20 sparse_b=$sb', round(d['ms_per_step'],3), d['config']['msm'].get('sparse_b'), {k: round(v,2) for k,v in d['stages_ms_per_step'].items() if v})"
done
timeout 600 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 --no-pmc > $O/bench_poseidon20.json 2> $O/bench_poseidon20.err
python -c "
import json; d=json.loads(open('$O/bench_poseidon20.json').read().strip().splitlines()[-1]); print('poseidon 2^20 default', round(d['ms_per_step'],3), d['value'], d['parity'], d['config']['msm'].get('sparse_b'), d['cpu_baseline']['value'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-log2 0 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d.get('clock_mhz'), d.get('power_w'))"
