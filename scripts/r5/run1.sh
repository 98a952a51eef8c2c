#!/bin/bash
# round 5, GPU call 1: the real Poseidon chain (configs[4]) on the GPU -- tests + bench line
O=gpurun_out/r5_1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "poseidon or more_wires" --durations=5 > $O/pytest_poseidon.log 2>&1; echo "pytest rc=$?" >> $O/pytest_poseidon.log
tail -5 $O/pytest_poseidon.log
timeout 600 python bench.py --workload poseidon --log2 20 --steps 20 --warmup 3 > $O/bench_poseidon20.json 2> $O/bench_poseidon20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_1/bench_poseidon20.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["parity"], d["stages_ms_per_step"])
print(d["cpu_baseline"])
PY
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-log2 0 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r5_1/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms_per_step'])"
