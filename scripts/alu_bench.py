"""integer-ALU ceilings on the GPU (v_mad_u64_u32, Fq mul, G1 madd) -> gpurun_out/alu_bench.json

Needs a measurement build of the library: make -C circom_compat_amd/csrc EXTRA=-DG16_DEBUG_ABI OUT=../libg16_amd_dbg.so
BUILD=../../build/hip_dbg, then G16_AMD_LIB=circom_compat_amd/libg16_amd_dbg.so python scripts/alu_bench.py
(the product library does not export g16_debug_alu_bench)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from circom_compat_amd import _binding

lib = _binding.load()
out = {}
for kind, name, blocks, iters in ((1, "v_mad_u64_u32", 4096, 4096), (0, "fq_mul", 4096, 512), (2, "g1_madd", 8192, 64),
                                  (3, "g1_madd_lazy29", 8192, 64), (4, "g2_madd_lazy29", 8192, 32)):
    sec, ops = C.c_double(), C.c_double()
    st = lib.g16_debug_alu_bench(0, kind, blocks, iters, C.byref(sec), C.byref(ops))
    assert st == 0, st
    out[name] = {"seconds": sec.value, "ops": ops.value, "ops_per_s": ops.value / sec.value}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "alu_bench.json"), "w"), indent=1)
