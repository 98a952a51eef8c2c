cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "in_library_multi_device_prover_large" --durations=4 ) 2>&1 | tail -12
