#!/bin/bash
# round 5, GPU call 15: kernel trace of the configs[4] line (Poseidon chain 2^20): what the filtered view costs
# (k_bucket_count<true> / k_bucket_scatter<true>) and what the B2 launch takes over it
O=gpurun_out/r5_15; mkdir -p $O; export TMPDIR=/tmp
R=$PWD; rm -rf /tmp/prof_p; cd /tmp
G16_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o kt -- python $R/bench.py --workload poseidon --log2 20 --steps 10 --warmup 2 --cpu-log2 0 --no-pmc > $R/$O/kt.log 2>&1
cd $R; DB=$(find /tmp/prof_p -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB > $O/r05_poseidon20_kernel_stats.txt 2>&1; head -30 $O/r05_poseidon20_kernel_stats.txt | cut -c1-170
