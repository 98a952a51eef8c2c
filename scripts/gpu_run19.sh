cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
free -g | head -2; nproc
( time timeout 1500 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "headline_sizes and 24" --durations=3 ) 2>&1 | tail -12
