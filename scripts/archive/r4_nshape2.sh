#!/bin/bash
# round 4: the bench N > 1 shapes as GPU tests (after restoring bench.py's per-process helpers)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4_nshape2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bench_shapes.py -m gpu -x -q --durations=5 > $O/bench_shapes.log 2>&1; echo "rc=$?"; tail -15 $O/bench_shapes.log | cut -c1-300
