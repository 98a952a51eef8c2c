#!/bin/bash
# Clock / power trace (rocm-smi samples every ~0.3 s) while (a) full 2^22 proofs and (b) the register-only
# mixed-addition micro-benchmark run: evidence for the DVFS statement of DESIGN.md section 5.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
sample() {  # $1 = label, $2 = pid to watch
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e "s/^/$1 /"
    sleep 0.3
  done
}
G16_BENCH_NO_PIPELINE=1 python bench.py --steps 300 --warmup 2 --cpu-log2 0 > gpurun_out/clock_trace_bench.json 2>/dev/null &
sample prove $!
python - <<'PY' &
import ctypes as C, sys
sys.path.insert(0, ".")
from circom_compat_amd import _binding
lib = _binding.load()
for kind, blocks, iters in ((3, 8192, 64), (4, 8192, 32)):
    for _ in range(600 if kind == 3 else 400):
        s, o = C.c_double(), C.c_double()
        lib.g16_debug_alu_bench(0, kind, blocks, iters, C.byref(s), C.byref(o))
    print("kind", kind, o.value / s.value, flush=True)
PY
sample alu $!
