#!/bin/bash
# round 3, GPU run 2: rank timelines (bucket-range sharding) + the RCCL world-1 test's failure output
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "rccl or poseidon or config5 or libsnark" > gpurun_out/r3_run2_pytest.log 2>&1
tail -40 gpurun_out/r3_run2_pytest.log
cd /tmp
for cfg in "22 8 buckets 4" "24 8 buckets 4"; do
  set -- $cfg
  rm -rf /tmp/prof_$1
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$1 -o trace -- python /root/repo/scripts/dist_rank_trace.py $1 $2 $3 $4 3 > /root/repo/gpurun_out/r3_trace_$1.log 2>&1
  db=$(find /tmp/prof_$1 -name "*.db" | head -1)
  python /root/repo/scripts/rocpd_timeline.py $db 160 > /root/repo/gpurun_out/r3_rank8_timeline_k$1_buckets.txt 2>&1
  tail -3 /root/repo/gpurun_out/r3_trace_$1.log
done
