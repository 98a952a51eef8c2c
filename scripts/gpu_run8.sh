# round-2 profile set: kernel trace + PMC passes over the dominant kernel + bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_kt -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-log2 0 > $GRAFT_REPO_ROOT/gpurun_out/r02_kt.log 2>&1 )
DB=$(find gpurun_out/r02_kt -name "*.db" | head -1); python scripts/rocpd_stats.py $DB > gpurun_out/r02_k22_kernel_stats.txt; head -12 gpurun_out/r02_k22_kernel_stats.txt
find gpurun_out/r02_kt -name "*stats*" | head; 
bash scripts/pmc_passes.sh 22 r02_pmc k_bucket_accumulate > gpurun_out/r02_pmc.log 2>&1; tail -3 gpurun_out/r02_pmc.log
python scripts/pmc_traffic.py gpurun_out/r02_pmc 22 gpurun_out/r02_pmc_traffic.json
python scripts/pmc_summary.py gpurun_out/r02_pmc > gpurun_out/r02_pmc_k22_accumulate.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; head -c 400 gpurun_out/r02_bench_default.json; echo
python bench.py --mode parts --log2 20 --steps 5 --cpu-log2 0 > gpurun_out/r02_bench_parts_k20.json 2>/dev/null
python bench.py --workload dense-skewed --log2 20 --steps 10 > gpurun_out/r02_bench_dense20.json 2>/dev/null
python bench.py --log2 20 --steps 10 > gpurun_out/r02_bench_chain20.json 2>/dev/null
python bench.py --workload complex-circuit --steps 20 > gpurun_out/r02_bench_complex.json 2>/dev/null
