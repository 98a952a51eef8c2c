cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_default.json')); print(d['value'], d['ms_per_step'], d['value_pcie_inclusive'], d['cpu_baseline'], d['parity'])"
python bench.py --workload complex-circuit --steps 20 > gpurun_out/r02_bench_complex.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_complex.json')); print(d['ms_per_step'], d['cpu_baseline']['sample'][:90], d['parity'])"
python bench.py --workload dense-skewed --log2 20 --steps 10 > gpurun_out/r02_bench_dense20.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_dense20.json')); print(d['ms_per_step'], d['cpu_baseline']['sample'][:90], d['cpu_baseline']['value'], d['parity'])"
