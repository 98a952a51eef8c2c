#!/bin/bash
# round 4, GPU run 5: (a) the multi-device tests five times over (the self-test's clean-up race fix),
# (b) full GPU suite, (c) same-box A/B: non-temporal point gathers (variant library) vs the product
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
echo "== (a) multi-device tests x5"
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_kernels.py tests/test_gpu_large.py -m gpu -x -q -k "self_test or in_library_multi_device_prover or multi_device_prover_on_the_reference" > $O/multi_$i.log 2>&1; echo "round $i rc=$? $(tail -1 $O/multi_$i.log)"
done
echo "== (b) full GPU suite"
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $O/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/r04_pytest_gpu.log
echo "== (c) non-temporal gathers"
REPS=3 bash scripts/ab.sh $O/nt "--steps 20 --warmup 3" "-" "G16_AMD_LIB=scripts/variants/libg16_nt_gather.so"
