cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
nproc > gpurun_out/r02_nproc.txt; rocm-smi --showclocks 2>/dev/null | head -20 >> gpurun_out/r02_nproc.txt
timeout 1200 python -m pytest tests -m gpu -x -q --durations=20 > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r02_pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc=$?"
./scripts/ubench/instr_rates > gpurun_out/r02_instr_rates.txt 2>&1
timeout 300 python bench.py --mode parts --log2 20 --steps 5 --cpu-log2 0 > gpurun_out/r02_bench_parts_k20.json 2> gpurun_out/r02_bench_parts_k20.err; echo "parts rc=$?"
timeout 600 python bench.py --workload dense-skewed --log2 20 --steps 5 > gpurun_out/r02_bench_dense20.json 2> gpurun_out/r02_bench_dense20.err; echo "dense rc=$?"
timeout 300 python bench.py --workload complex-circuit --steps 10 > gpurun_out/r02_bench_complex.json 2> gpurun_out/r02_bench_complex.err; echo "complex rc=$?"
G16_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 4 --log2 18 --steps 3 --cpu-log2 14 > gpurun_out/r02_bench_inlib4_onegpu.json 2> gpurun_out/r02_bench_inlib4_onegpu.err; echo "inlib rc=$?"
cat gpurun_out/r02_bench_default.json | head -c 3000
