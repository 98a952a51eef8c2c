"""ctypes binding of the C ABI in include/g16_amd.h + include/g16_loaders.h.

This is the same stub a Rust maintainer would write as `extern "C"`; Python is
only the harness language of tests/ and bench.py.  There is no CPU fallback: `load()` raises when
the HIP library has not been built, and ctx creation fails with G16_ERR_NO_DEVICE without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libg16_amd.so")

G16_OK, G16_ERR_INVALID, G16_ERR_DOMAIN_TOO_LARGE, G16_ERR_HIP, G16_ERR_NO_DEVICE, G16_ERR_IO, \
    G16_ERR_INTERNAL = range(7)
G16_PROOF_BYTES = 256
G16_PARTIAL_BYTES = 1024
G16_N_STAGES = 10
QUERY_A, QUERY_B1, QUERY_L, QUERY_H = 0, 1, 2, 3

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class G16Error(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"g16 status {status}: {message}")
        self.status = status
        self.message = message


class SynthesisError(G16Error):
    """PolynomialDegreeTooLarge (reference src/circom/qap.rs:31,66)."""


class SerializationError(G16Error):
    """ark_serialize::SerializationError stand-in for loader failures."""


class Csr(C.Structure):
    _fields_ = [("row_ptr", _u32p), ("col", _u32p), ("coeff", _u64p), ("nnz", C.c_uint64)]


class KeyDesc(C.Structure):
    _fields_ = [("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32),
                ("a_query", C.c_void_p), ("b_g1_query", C.c_void_p), ("b_g2_query", C.c_void_p),
                ("l_query", C.c_void_p), ("h_query", C.c_void_p),
                ("alpha_g1", C.c_uint8 * 64), ("beta_g1", C.c_uint8 * 64),
                ("delta_g1", C.c_uint8 * 64), ("beta_g2", C.c_uint8 * 128),
                ("delta_g2", C.c_uint8 * 128)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("window_bits", C.c_int), ("planes", C.c_int), ("dist_wm", C.c_int),
                ("reduction", C.c_int), ("shard", C.c_int), ("fixed_tables", C.c_int)]


SHARD_AUTO, SHARD_POINTS, SHARD_BUCKETS = 0, 1, 2


class VkDesc(C.Structure):
    _fields_ = [("alpha_g1", C.c_uint8 * 64), ("beta_g2", C.c_uint8 * 128), ("gamma_g2", C.c_uint8 * 128),
                ("delta_g2", C.c_uint8 * 128), ("ic", C.c_void_p), ("ic_count", C.c_uint32)]


class ZkeyHeader(C.Structure):
    _fields_ = [("n8q", C.c_uint32), ("n8r", C.c_uint32), ("q", C.c_uint8 * 32),
                ("r", C.c_uint8 * 32), ("n_vars", C.c_uint32), ("n_public", C.c_uint32),
                ("domain_size", C.c_uint32), ("power", C.c_uint32),
                ("alpha_g1", C.c_uint8 * 64), ("beta_g1", C.c_uint8 * 64),
                ("beta_g2", C.c_uint8 * 128), ("gamma_g2", C.c_uint8 * 128),
                ("delta_g1", C.c_uint8 * 64), ("delta_g2", C.c_uint8 * 128)]


class Matrices(C.Structure):
    _fields_ = [("num_instance_variables", C.c_uint32), ("num_witness_variables", C.c_uint32),
                ("num_constraints", C.c_uint32), ("a_num_non_zero", C.c_uint64),
                ("b_num_non_zero", C.c_uint64), ("a", Csr), ("b", Csr)]


class R1csHeader(C.Structure):
    _fields_ = [("version", C.c_uint32), ("field_size", C.c_uint32), ("prime", C.c_uint8 * 32),
                ("n_wires", C.c_uint32), ("n_pub_out", C.c_uint32), ("n_pub_in", C.c_uint32),
                ("n_prv_in", C.c_uint32), ("n_labels", C.c_uint64), ("n_constraints", C.c_uint32),
                ("num_inputs", C.c_uint32), ("num_aux", C.c_uint32), ("num_variables", C.c_uint32)]


# every symbol include/*.h declares; tests assert the built library exports all of them
ABI_SYMBOLS = [
    "g16_ctx_create", "g16_ctx_create_sibling", "g16_ctx_destroy", "g16_last_error", "g16_witness_map", "g16_msm_g1",
    "g16_msm_g2", "g16_prove", "g16_prove_dev", "g16_prove_partial", "g16_prove_partial_dev",
    "g16_prove_finish", "g16_dist_exchange_bytes", "g16_prove_dist_phase1", "g16_prove_dist_phase2",
    "g16_prove_dist_phase3", "g16_set_profiling", "g16_stage_times", "g16_stage_name", "g16_ctx_info", "g16_multi_links",
    "g16_witness_buffer", "g16_witness_upload", "g16_witness_host_buffer", "g16_ctx_create_multi", "g16_dist_set_exchange_stream",
    "g16_partial_buffer", "g16_gather_buffer", "g16_prove_finish_dev", "g16_witness_map_dev", "g16_msm_g1_dev",
    "g16_msm_g2_dev", "g16_verify_batch", "g16_fft_in_place", "g16_dist_attach_rccl", "g16_dist_rccl_ranks", "g16_prove_dist", "g16_check_satisfied", "g16_zkey_write",
    "g16_setup_create", "g16_setup_create_ex", "g16_setup_destroy", "g16_setup_key",
    "g16_loader_last_error", "g16_zkey_open", "g16_zkey_open_mem", "g16_zkey_close",
    "g16_zkey_header_get", "g16_zkey_key", "g16_zkey_ic", "g16_zkey_matrices", "g16_r1cs_open",
    "g16_r1cs_open_mem", "g16_r1cs_close", "g16_r1cs_header_get", "g16_r1cs_matrices",
    "g16_r1cs_wire_mapping", "g16_wtns_read", "g16_wtns_read_mem", "g16_free",
    "g16_fr_from_canonical", "g16_fr_to_canonical",
]


class Library:
    """Loaded libg16_amd.so with typed entry points."""

    def __init__(self, path: Optional[str] = None):
        path = path or os.environ.get("G16_AMD_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C circom_compat_amd/csrc). "
                "There is no CPU fallback.")
        self.path = path
        # PyTorch-ROCm wheels carry their own HIP runtime.  If libg16_amd.so (linked against
        # /opt/rocm's libamdhip64) is loaded first, a later `import torch` ends up with a second
        # runtime and reports "No HIP GPUs are available".  Host frameworks that use torch for
        # device buffers / RCCL therefore get its runtime loaded first; G16_NO_TORCH_PRELOAD=1 skips it.
        if not os.environ.get("G16_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = self.L = C.CDLL(path)
        vp = C.c_void_p
        sig = {
            "g16_ctx_create": (C.c_int, [C.POINTER(KeyDesc), C.POINTER(Csr), C.POINTER(Csr),
                                         C.c_uint32, C.POINTER(Options), C.POINTER(vp)]),
            "g16_ctx_create_sibling": (C.c_int, [vp, C.POINTER(KeyDesc), C.POINTER(Csr), C.POINTER(Csr),
                                                 C.c_uint32, C.POINTER(Options), C.POINTER(vp)]),
            "g16_ctx_destroy": (None, [vp]),
            "g16_last_error": (C.c_char_p, [vp]),
            "g16_witness_map": (C.c_int, [vp, vp, C.c_size_t, vp]),
            "g16_msm_g1": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp]),
            "g16_msm_g2": (C.c_int, [vp, vp, C.c_size_t, vp]),
            "g16_prove": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_prove_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_prove_partial": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_prove_partial_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_prove_finish": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
            "g16_dist_exchange_bytes": (C.c_size_t, [vp]),
            "g16_prove_dist_phase1": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_prove_dist_phase2": (C.c_int, [vp, vp, vp]),
            "g16_prove_dist_phase3": (C.c_int, [vp, vp, vp]),
            "g16_set_profiling": (C.c_int, [vp, C.c_int]),
            "g16_stage_times": (C.c_int, [vp, C.POINTER(C.c_float), _u32p]),
            "g16_stage_name": (C.c_char_p, [C.c_int]),
            "g16_multi_links": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_uint64)]),
            "g16_ctx_info": (C.c_int, [vp, _u32p]),
            "g16_witness_buffer": (vp, [vp]),
            "g16_witness_host_buffer": (vp, [vp]),
            "g16_verify_batch": (C.c_int, [C.c_int, C.POINTER(VkDesc), vp, vp, C.c_uint32, vp]),
            "g16_witness_upload": (C.c_int, [vp, vp, C.c_size_t]),
            "g16_witness_map_dev": (C.c_int, [vp, vp, C.c_size_t, vp]),
            "g16_msm_g1_dev": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp]),
            "g16_msm_g2_dev": (C.c_int, [vp, vp, C.c_size_t, vp]),
            "g16_ctx_create_multi": (C.c_int, [C.POINTER(KeyDesc), C.POINTER(Csr), C.POINTER(Csr), C.c_uint32,
                                               C.POINTER(C.c_int), C.c_int, C.POINTER(Options), C.POINTER(vp)]),
            "g16_dist_set_exchange_stream": (C.c_int, [vp, vp, C.c_int]),
            "g16_partial_buffer": (vp, [vp]),
            "g16_gather_buffer": (vp, [vp]),
            "g16_prove_finish_dev": (C.c_int, [vp, vp, vp, vp]),
            "g16_fft_in_place": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int]),
            "g16_dist_attach_rccl": (C.c_int, [vp, vp]),
            "g16_dist_rccl_ranks": (C.c_int, [vp]),
            "g16_prove_dist": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
            "g16_check_satisfied": (C.c_int, [C.c_int, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr),
                                              C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_int64)]),
            "g16_zkey_write": (C.c_int, [C.c_char_p, C.POINTER(KeyDesc), vp, vp, C.POINTER(Csr),
                                         C.POINTER(Csr), C.c_uint32]),
            "g16_loader_last_error": (C.c_char_p, []),
            "g16_zkey_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
            "g16_zkey_open_mem": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
            "g16_zkey_close": (None, [vp]),
            "g16_zkey_header_get": (C.c_int, [vp, C.POINTER(ZkeyHeader)]),
            "g16_zkey_key": (C.c_int, [vp, C.POINTER(KeyDesc)]),
            "g16_zkey_ic": (vp, [vp, _u32p]),
            "g16_zkey_matrices": (C.c_int, [vp, C.POINTER(Matrices)]),
            "g16_r1cs_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
            "g16_r1cs_open_mem": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
            "g16_r1cs_close": (None, [vp]),
            "g16_r1cs_header_get": (C.c_int, [vp, C.POINTER(R1csHeader)]),
            "g16_r1cs_matrices": (C.c_int, [vp, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr)]),
            "g16_r1cs_wire_mapping": (vp, [vp, _u32p]),
            "g16_wtns_read": (C.c_int, [C.c_char_p, C.POINTER(vp), _u32p]),
            "g16_wtns_read_mem": (C.c_int, [vp, C.c_size_t, C.POINTER(vp), _u32p]),
            "g16_free": (None, [vp]),
            "g16_fr_from_canonical": (C.c_int, [vp, vp, C.c_size_t]),
            "g16_fr_to_canonical": (C.c_int, [vp, vp, C.c_size_t]),
            "g16_setup_create": (C.c_int, [C.c_int, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr),
                                           C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(vp)]),
            "g16_setup_create_ex": (C.c_int, [C.c_int, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr),
                                           C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_int, C.POINTER(vp)]),
            "g16_setup_destroy": (None, [vp]),
            "g16_setup_key": (C.c_int, [vp, C.POINTER(KeyDesc), C.POINTER(vp), _u32p, vp]),
        }
        # measurement builds only (make EXTRA=-DG16_DEBUG_ABI; include/g16_amd.h): not an ABI symbol, never "missing"
        optional = {"g16_debug_alu_bench": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                                      C.POINTER(C.c_double), C.POINTER(C.c_double)])}
        for name, (res, args) in optional.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
                setattr(self, name, fn)
        self.missing = []
        for name, (res, args) in sig.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, status, ctx=None, loader=False):
        if status == G16_OK:
            return
        if loader:
            msg = self.g16_loader_last_error().decode()
            raise SerializationError(status, msg)
        msg = self.g16_last_error(ctx).decode()
        if status == G16_ERR_DOMAIN_TOO_LARGE:
            raise SynthesisError(status, msg)
        raise G16Error(status, msg)


_default: Optional[Library] = None


def load(path: Optional[str] = None) -> Library:
    """The product library (cached).  Raises ImportError when it has not been built."""
    global _default
    if path is not None:
        return Library(path)
    if _default is None:
        _default = Library()
    return _default
