#!/bin/bash
# round 5, GPU call 3: fixed-base tables for small keys (parity + latency), clock sampler on the right card
O=gpurun_out/r5_5; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "fixed_base" --durations=8 > $O/pytest_tables.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tables.log
tail -12 $O/pytest_tables.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --workload complex-circuit --steps 200 --warmup 20 --no-pmc > $O/bench_complex.json 2> $O/bench_complex.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_5/bench_complex.json").read().strip().splitlines()[-1])
print("complex-circuit", d["ms_per_step"], d["ms_per_step_pcie_inclusive"], d["parity"], d["stages_ms_per_step"], d["config"]["msm"], d.get("value_pipelined"), d.get("clock_mhz"), d.get("power_w"), d["setup_s"])
print(d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, d["clock"]["source"] if d.get("clock") else None)
PY
G16_NO_OVERLAP=1 timeout 600 python bench.py --workload complex-circuit --steps 50 --warmup 5 --no-pmc --cpu-log2 0 > $O/bench_complex_noov.json 2> $O/bench_complex_noov.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_5/bench_complex_noov.json").read().strip().splitlines()[-1])
print("complex-circuit one stream", d["ms_per_step"], d["stages_ms_per_step"])
PY
timeout 900 python scripts/bench_sweep.py 50 > $O/bench_sweep.txt 2> $O/bench_sweep.err; head -9 $O/bench_sweep.txt
for k in 10 12 14; do
  timeout 300 python bench.py --log2 $k --steps 100 --warmup 10 --no-pmc --cpu-log2 0 > $O/b$k.json 2> $O/b$k.err
  python -c "
import json; d=json.loads(open('$O/b$k.json').read().strip().splitlines()[-1]); print('chain 2^$k', d['ms_per_step'], d['config']['msm']['fixed_tables'], d['setup_s'])"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload complex-circuit --steps 20 --warmup 2 --no-pmc --cpu-log2 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls $O/prof* | head; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -30 {}
timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-log2 0 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d.get('clock_mhz'), d.get('power_w'), d.get('clock'))"
