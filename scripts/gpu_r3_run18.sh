#!/bin/bash
# round 3, GPU run 18: the 2^25 capacity-point test with the byte comparison against the CPU restatement
cd /root/repo
mkdir -p gpurun_out/run18
G16_TEST_2P25_BYTES=1 timeout 1200 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k capacity_point --durations=3 2>&1 | tail -12 | tee gpurun_out/run18/r03_pytest_2p25_bytes.log
