# profile set of the final round-2 binary (A|B1 interleaved pair): GPU tests, kernel trace, PMC passes, bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_kt2 -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-log2 0 > $GRAFT_REPO_ROOT/gpurun_out/r02_kt2.log 2>&1 )
DB=$(find gpurun_out/r02_kt2 -name "*.db" | head -1); python scripts/rocpd_stats.py $DB > gpurun_out/r02_k22_kernel_stats.txt; head -14 gpurun_out/r02_k22_kernel_stats.txt
bash scripts/pmc_passes.sh 22 r02_pmc2 k_bucket_accumulate > gpurun_out/r02_pmc2.log 2>&1; tail -3 gpurun_out/r02_pmc2.log
python scripts/pmc_traffic.py gpurun_out/r02_pmc2 22 gpurun_out/r02_pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/r02_pmc2 > gpurun_out/r02_pmc_k22_accumulate.txt
cp gpurun_out/r02_pmc_traffic.json profiles/pmc_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_final.json')); print(d['value'], d['ms_per_step'], d['value_pcie_inclusive'], json.dumps(d['roofline']), d['cpu_baseline']['value'], d['parity'])"
G16_NO_PAIR_AB=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate arrays:', d['ms_per_step'], d['stages_ms_per_step'])"
