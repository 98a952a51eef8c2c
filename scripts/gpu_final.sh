# end-of-round validation on a fresh box: GPU tests, smoke, the driver's bench command, N>1 launch shapes (functional)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_final.json')); print(d['value'], d['ms_per_step'], d['value_pcie_inclusive'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity'])"
G16_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --log2 18 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('torchrun inlib N=2', d['ms_per_step'], d['parity'])"
