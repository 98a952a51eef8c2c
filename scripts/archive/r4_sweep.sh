#!/bin/bash
# round 4: the reference bench's sweep and the batch-verifier throughput on the final binary
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4_sweep; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/bench_sweep.py 20 > $O/r04_bench_sweep.txt 2> $O/sweep.err; echo "rc=$?"; cat $O/r04_bench_sweep.txt | cut -c1-200
timeout 600 python scripts/verify_bench.py > $O/r04_verify_bench.txt 2> $O/verify.err; echo "rc=$?"; cat $O/r04_verify_bench.txt | cut -c1-200
