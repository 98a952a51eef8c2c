"""circom_compat_amd -- MI355X-native Groth16 (BN254) proving path for Circom circuits.

Host-side mirror (Python harness flavour) of the reference's surface for the proving path:

    reference (ark-circom 0.5)                              here
    ------------------------------------------------------  -----------------------------------
    read_zkey(reader) -> (ProvingKey, ConstraintMatrices)   read_zkey(path | bytes)
        src/zkey.rs:53-60
    R1CSFile::new(reader), R1CS::from(file)                 R1CSFile(path | bytes), R1CS.from_file
        src/circom/r1cs_reader.rs:26-39,54-146
    CircomCircuit{r1cs, witness}.get_public_inputs()        CircomCircuit(...).get_public_inputs()
        src/circom/circuit.rs:12-26
    CircomReduction::witness_map_from_matrices              CircomReduction.witness_map_from_matrices
        src/circom/qap.rs:23-88
    Groth16::<Bn254,CircomReduction>::                      Groth16.create_proof_with_reduction_and_matrices
        create_proof_with_reduction_and_matrices            (same argument order)
        benches/groth16.rs:52-60, src/zkey.rs:903-911
    Groth16::prove(&pk, circuit, rng)  src/zkey.rs:866      Groth16.prove(pk, matrices, circuit, rng)

All arithmetic happens in libg16_amd.so (hand-written HIP for gfx950) behind the C ABI of
include/g16_amd.h.  Witness generation (circom WASM), Ethereum helpers and the arkworks
ConstraintSystem layer are out of scope.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _binding as B
from ._binding import G16Error, SerializationError, SynthesisError  # noqa: F401

__all__ = ["read_zkey", "R1CSFile", "R1CS", "CircomCircuit", "CircomBuilder", "CircomReduction", "LibsnarkReduction", "Groth16",
           "Prover", "ProvingKey", "VerifyingKey", "ConstraintMatrices", "Proof", "G16Error",
           "SynthesisError", "SerializationError", "fr_from_ints", "fr_to_ints", "read_wtns",
           "trapdoor_setup", "Csr", "write_zkey", "device_tensor", "verify_batch"]

FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fr_from_ints(values: Sequence[int], lib: Optional[B.Library] = None) -> np.ndarray:
    """canonical ints -> (n, 4) uint64 Montgomery limbs (the in-memory form of ark_bn254::Fr)."""
    lib = lib or B.load()
    raw = b"".join((int(v) % FR_MODULUS).to_bytes(32, "little") for v in values)
    src = np.frombuffer(raw, dtype=np.uint8)
    out = np.empty((len(values), 4), dtype=np.uint64)
    lib.check(lib.g16_fr_from_canonical(_np_ptr(src), _np_ptr(out), len(values)), loader=True)
    return out


def fr_to_ints(limbs: np.ndarray, lib: Optional[B.Library] = None) -> List[int]:
    lib = lib or B.load()
    limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4)
    out = np.empty(limbs.shape[0] * 32, dtype=np.uint8)
    lib.check(lib.g16_fr_to_canonical(_np_ptr(limbs), _np_ptr(out), limbs.shape[0]), loader=True)
    b = out.tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _as_fr(x, lib) -> np.ndarray:
    """accept Montgomery (n,4) uint64 arrays or sequences of canonical ints"""
    if isinstance(x, np.ndarray) and x.dtype == np.uint64:
        return np.ascontiguousarray(x).reshape(-1, 4)
    return fr_from_ints(list(x), lib)


class _RawDeviceBytes:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2}


def device_tensor(ptr: int, nbytes: int, device=None):
    """uint8 torch view of library-owned device memory (g16_partial_buffer / g16_gather_buffer), so
    that a host framework can hand it to its collectives (RCCL all_gather) without a host copy.
    Plumbing only: torch is imported here, never by the proving path."""
    import torch
    return torch.as_tensor(_RawDeviceBytes(ptr, nbytes), device=device or "cuda")


class Csr:
    """row-major sparse rows of (coeff, index): ConstraintMatrices::{a,b} (src/zkey.rs:165-194)"""

    def __init__(self, row_ptr, col, coeff):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.coeff = np.ascontiguousarray(coeff, dtype=np.uint64).reshape(-1, 4)
        assert self.col.shape[0] == self.coeff.shape[0] == int(self.row_ptr[-1])

    @staticmethod
    def from_c(c: B.Csr, nrows: int) -> "Csr":
        nnz = int(c.nnz)
        rp = np.ctypeslib.as_array(c.row_ptr, shape=(nrows + 1,)).copy()
        col = np.ctypeslib.as_array(c.col, shape=(max(nnz, 1),))[:nnz].copy()
        co = np.ctypeslib.as_array(c.coeff, shape=(max(nnz, 1) * 4,))[:nnz * 4].copy()
        return Csr(rp, col, co)

    @staticmethod
    def from_rows(rows, lib=None) -> "Csr":
        """rows: list of lists of (coeff_int, index)"""
        rp = [0]
        col, vals = [], []
        for row in rows:
            for cf, idx in row:
                col.append(idx)
                vals.append(cf)
            rp.append(len(col))
        return Csr(rp, col, fr_from_ints(vals, lib) if vals else np.zeros((0, 4), np.uint64))

    def to_c(self) -> B.Csr:
        c = B.Csr()
        c.row_ptr = self.row_ptr.ctypes.data_as(C.POINTER(C.c_uint32))
        c.col = self.col.ctypes.data_as(C.POINTER(C.c_uint32))
        c.coeff = self.coeff.ctypes.data_as(C.POINTER(C.c_uint64))
        c.nnz = self.col.shape[0]
        return c

    @property
    def num_rows(self):
        return self.row_ptr.shape[0] - 1


class ConstraintMatrices:
    """ark_relations::r1cs::ConstraintMatrices as read_zkey fills it (src/zkey.rs:179-193)."""

    def __init__(self, num_instance_variables, num_witness_variables, num_constraints, a: Csr,
                 b: Csr):
        self.num_instance_variables = num_instance_variables
        self.num_witness_variables = num_witness_variables
        self.num_constraints = num_constraints
        self.a, self.b = a, b
        self.a_num_non_zero = a.col.shape[0]
        self.b_num_non_zero = b.col.shape[0]
        self.c_num_non_zero = 0  # src/zkey.rs:188-192: c is empty


class VerifyingKey:
    def __init__(self, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1):
        self.alpha_g1, self.beta_g2, self.gamma_g2, self.delta_g2 = alpha_g1, beta_g2, gamma_g2, delta_g2
        self.gamma_abc_g1 = gamma_abc_g1  # (p+1, 64) uint8


class ProvingKey:
    """ark_groth16::ProvingKey<Bn254> in packed form (points: Montgomery x|y bytes)."""

    def __init__(self, n_vars, n_public, domain_size, vk: VerifyingKey, beta_g1, delta_g1, a_query,
                 b_g1_query, b_g2_query, l_query, h_query, keepalive=None):
        self.n_vars, self.n_public, self.domain_size = n_vars, n_public, domain_size
        self.vk, self.beta_g1, self.delta_g1 = vk, beta_g1, delta_g1
        self.a_query, self.b_g1_query, self.b_g2_query = a_query, b_g1_query, b_g2_query
        self.l_query, self.h_query = l_query, h_query
        self._keepalive = keepalive
        self._prover = None

    def to_c(self) -> B.KeyDesc:
        k = B.KeyDesc()
        k.n_vars, k.n_public, k.domain_size = self.n_vars, self.n_public, self.domain_size
        for name in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
            arr = getattr(self, name)
            setattr(k, name, arr.ctypes.data if arr is not None else None)
        C.memmove(k.alpha_g1, bytes(self.vk.alpha_g1), 64)
        C.memmove(k.beta_g1, bytes(self.beta_g1), 64)
        C.memmove(k.delta_g1, bytes(self.delta_g1), 64)
        C.memmove(k.beta_g2, bytes(self.vk.beta_g2), 128)
        C.memmove(k.delta_g2, bytes(self.vk.delta_g2), 128)
        return k


class _Handle:
    def __init__(self, lib, ptr, closer):
        self.lib, self.ptr, self._closer = lib, ptr, closer

    def __del__(self):
        try:
            if self.ptr:
                self._closer(self.ptr)
        except Exception:
            pass
        self.ptr = None


def read_zkey(src, lib: Optional[B.Library] = None) -> Tuple[ProvingKey, ConstraintMatrices]:
    """read_zkey (reference src/zkey.rs:53-60): snarkjs .zkey -> (ProvingKey, ConstraintMatrices)."""
    lib = lib or B.load()
    h = C.c_void_p()
    if isinstance(src, (bytes, bytearray, memoryview)):
        buf = np.frombuffer(bytes(src), dtype=np.uint8)
        lib.check(lib.g16_zkey_open_mem(_np_ptr(buf), buf.shape[0], C.byref(h)), loader=True)
    else:
        lib.check(lib.g16_zkey_open(os.fsencode(src), C.byref(h)), loader=True)
    handle = _Handle(lib, h, lib.g16_zkey_close)
    hdr = B.ZkeyHeader()
    lib.check(lib.g16_zkey_header_get(h, C.byref(hdr)), loader=True)
    kd = B.KeyDesc()
    lib.check(lib.g16_zkey_key(h, C.byref(kd)), loader=True)
    N, p, n = hdr.n_vars, hdr.n_public, hdr.domain_size

    def view(ptr, count, width):
        if count == 0:
            return np.zeros((0, width), dtype=np.uint8)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * width,))
        return arr.reshape(count, width)

    cnt = C.c_uint32()
    icp = lib.g16_zkey_ic(h, C.byref(cnt))
    vk = VerifyingKey(bytes(hdr.alpha_g1), bytes(hdr.beta_g2), bytes(hdr.gamma_g2),
                      bytes(hdr.delta_g2), view(icp, cnt.value, 64).copy())
    pk = ProvingKey(N, p, n, vk, bytes(hdr.beta_g1), bytes(hdr.delta_g1),
                    view(kd.a_query, N, 64), view(kd.b_g1_query, N, 64),
                    view(kd.b_g2_query, N, 128), view(kd.l_query, N - p - 1, 64),
                    view(kd.h_query, n, 64), keepalive=handle)
    m = B.Matrices()
    lib.check(lib.g16_zkey_matrices(h, C.byref(m)), loader=True)
    mats = ConstraintMatrices(m.num_instance_variables, m.num_witness_variables, m.num_constraints,
                              Csr.from_c(m.a, m.num_constraints), Csr.from_c(m.b, m.num_constraints))
    return pk, mats


class R1CSFile:
    """R1CSFile::new (reference src/circom/r1cs_reader.rs:54-146)."""

    def __init__(self, src, lib: Optional[B.Library] = None):
        lib = lib or B.load()
        h = C.c_void_p()
        if isinstance(src, (bytes, bytearray, memoryview)):
            buf = np.frombuffer(bytes(src), dtype=np.uint8)
            lib.check(lib.g16_r1cs_open_mem(_np_ptr(buf), buf.shape[0], C.byref(h)), loader=True)
        else:
            lib.check(lib.g16_r1cs_open(os.fsencode(src), C.byref(h)), loader=True)
        self._handle = _Handle(lib, h, lib.g16_r1cs_close)
        hdr = B.R1csHeader()
        lib.check(lib.g16_r1cs_header_get(h, C.byref(hdr)), loader=True)
        self.version = hdr.version
        self.header = hdr
        a, b, c = B.Csr(), B.Csr(), B.Csr()
        lib.check(lib.g16_r1cs_matrices(h, C.byref(a), C.byref(b), C.byref(c)), loader=True)
        nc = hdr.n_constraints
        self.a, self.b, self.c = Csr.from_c(a, nc), Csr.from_c(b, nc), Csr.from_c(c, nc)
        cnt = C.c_uint32()
        wm = lib.g16_r1cs_wire_mapping(h, C.byref(cnt))
        self.wire_mapping = list(np.ctypeslib.as_array(C.cast(wm, C.POINTER(C.c_uint64)),
                                                       shape=(cnt.value,)))


class R1CS:
    """R1CS::from(R1CSFile) (reference src/circom/r1cs_reader.rs:26-39)."""

    def __init__(self, file: R1CSFile):
        h = file.header
        self.num_inputs = 1 + h.n_pub_in + h.n_pub_out
        self.num_variables = h.n_wires
        self.num_aux = self.num_variables - self.num_inputs
        self.a, self.b, self.c = file.a, file.b, file.c
        self.num_constraints = h.n_constraints
        self.wire_mapping: Optional[List[int]] = [int(x) for x in file.wire_mapping]

    @staticmethod
    def from_file(src, lib=None) -> "R1CS":
        return R1CS(R1CSFile(src, lib))

    def matrices(self) -> ConstraintMatrices:
        """A and B in the ConstraintMatrices orientation the prover consumes."""
        return ConstraintMatrices(self.num_inputs, self.num_aux, self.num_constraints, self.a, self.b)


def write_zkey(path, pk: "ProvingKey", matrices: "ConstraintMatrices", lib: Optional[B.Library] = None):
    """snarkjs-format .zkey for a key held in packed arrays (inverse of read_zkey; format notes:
    reference src/zkey.rs:1-27).  Lets synthetic keys go through the loader path."""
    lib = lib or B.load()
    kd = pk.to_c()
    a, b = matrices.a.to_c(), matrices.b.to_c()
    ic = np.ascontiguousarray(pk.vk.gamma_abc_g1, dtype=np.uint8)
    g2 = np.frombuffer(bytes(pk.vk.gamma_g2), dtype=np.uint8)
    lib.check(lib.g16_zkey_write(os.fsencode(path), C.byref(kd), _np_ptr(ic), _np_ptr(g2), C.byref(a),
                                 C.byref(b), matrices.num_constraints), loader=True)


def read_wtns(src, lib: Optional[B.Library] = None) -> np.ndarray:
    """snarkjs .wtns -> (n, 4) uint64 Montgomery witness."""
    lib = lib or B.load()
    out = C.c_void_p()
    n = C.c_uint32()
    if isinstance(src, (bytes, bytearray, memoryview)):
        buf = np.frombuffer(bytes(src), dtype=np.uint8)
        lib.check(lib.g16_wtns_read_mem(_np_ptr(buf), buf.shape[0], C.byref(out), C.byref(n)),
                  loader=True)
    else:
        lib.check(lib.g16_wtns_read(os.fsencode(src), C.byref(out), C.byref(n)), loader=True)
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n.value * 4,)).copy()
    lib.g16_free(out)
    return arr.reshape(-1, 4)


class CircomCircuit:
    """CircomCircuit{r1cs, witness} (reference src/circom/circuit.rs:12-26).  The witness comes from
    outside (circom's WASM generator is out of scope): ints or Montgomery limbs."""

    def __init__(self, r1cs: R1CS, witness=None):
        self.r1cs = r1cs
        self.witness = witness

    def full_assignment(self):
        """The assignment CircomCircuit::generate_constraints builds (reference
        src/circom/circuit.rs:35-58): variable i takes witness[wire_mapping[i]] when the mapping is
        Some (what R1CS::from produces), witness[i] when it is None (what CircomBuilder::setup
        leaves, src/circom/builder.rs:84-85)."""
        w = self.witness
        m = getattr(self.r1cs, "wire_mapping", None)
        if w is None or m is None:
            return w
        n = self.r1cs.num_variables
        if isinstance(w, np.ndarray):
            return np.ascontiguousarray(w.reshape(-1, 4)[np.asarray(m[:n], dtype=np.int64)])
        return [w[m[i]] for i in range(n)]

    def first_unsatisfied(self, lib: Optional[B.Library] = None, device=0) -> int:
        """row index of the first constraint (A.w)(B.w) != C.w, or -1: the debug-build check of
        CircomBuilder::build (reference src/circom/builder.rs:101-114) as a GPU kernel"""
        lib = lib or B.load()
        w = _as_fr(self.full_assignment(), lib)
        a, b, c = self.r1cs.a.to_c(), self.r1cs.b.to_c(), self.r1cs.c.to_c()
        out = C.c_int64(-2)
        lib.check(lib.g16_check_satisfied(device, C.byref(a), C.byref(b), C.byref(c),
                                          self.r1cs.num_constraints, _np_ptr(w), w.shape[0],
                                          C.byref(out)))
        return int(out.value)

    def is_satisfied(self, lib: Optional[B.Library] = None) -> bool:
        return self.first_unsatisfied(lib) < 0

    def get_public_inputs(self):
        if self.witness is None:
            return None
        w = self.witness
        m = self.r1cs.wire_mapping
        if m is None:
            return [w[i] for i in range(1, self.r1cs.num_inputs)]
        return [w[m[i]] for i in range(1, self.r1cs.num_inputs)]


class CircomBuilder:
    """CircomBuilder (reference src/circom/builder.rs:60-117) minus the WASM witness calculator, which
    is out of scope: the witness comes from outside (snarkjs .wtns, JSON, another generator).  What
    it keeps is the part the proving path depends on: setup() / build() hand out circuits whose
    wire mapping is DISABLED (builder.rs:84-85), because circom witnesses are already in wire order
    -- so get_public_inputs(), the satisfiability check and Groth16.prove all read w[i]."""

    def __init__(self, r1cs: "R1CS"):
        self.r1cs = r1cs

    def setup(self) -> "CircomCircuit":
        import copy
        r = copy.copy(self.r1cs)
        r.wire_mapping = None  # "Disable the wire mapping"
        return CircomCircuit(r, None)

    def build(self, witness, sanity_check=False, lib=None) -> "CircomCircuit":
        c = self.setup()
        c.witness = witness
        if sanity_check:  # the debug_assert of builder.rs:101-114 as a kernel
            bad = c.first_unsatisfied(lib)
            if bad >= 0:
                raise G16Error(B.G16_ERR_INVALID, f"Unsatisfied constraint: {bad}")
        return c


class Proof:
    """ark_groth16::Proof<Bn254>{a, b, c} as packed affine bytes (Montgomery LE, zero = infinity)."""

    def __init__(self, raw: bytes):
        assert len(raw) == B.G16_PROOF_BYTES
        self.raw = bytes(raw)
        self.a, self.b, self.c = self.raw[:64], self.raw[64:192], self.raw[192:]

    def __eq__(self, o):
        return isinstance(o, Proof) and o.raw == self.raw

    def __repr__(self):
        return f"Proof({self.raw.hex()[:32]}...)"


DEFAULT_TABLES = 0     # g16_options.fixed_tables when Prover(tables=None): 0 = the library decides


class Prover:
    """Device-resident (pk, matrices): the state create_proof_with_reduction_and_matrices borrows
    on every call in the reference, uploaded and precomputed once here (g16_ctx_create)."""

    def __init__(self, pk: Optional[ProvingKey], matrices: ConstraintMatrices, device=0, rank=0,
                 world=1, window_bits=0, planes=0, lib: Optional[B.Library] = None,
                 n_vars: Optional[int] = None, dist_wm=False, reduction: str = "circom",
                 devices: Optional[Sequence[int]] = None, shard: str = "auto",
                 sibling_of: Optional["Prover"] = None, tables: Optional[int] = None):
        """devices=[d0, d1, ...]: ONE ctx sharded over several GPUs inside the library
        (g16_ctx_create_multi); prove() / prove_dev() are then used exactly as on one device.
        shard (world > 1 / devices): "points" = point-range MSM shards, "buckets" = every rank holds
        all points of the witness queries and 1/world of their sorted bucket list (H stays cut by
        point range), "auto" = points.
        sibling_of=prover: a second ctx on the same device that borrows `prover`'s point planes
        (g16_ctx_create_sibling): two threads, two proofs in flight.
        tables (g16_options.fixed_tables): None / 0 = automatic (small single-device keys prove through
        fixed-base tables), 1 = require, -1 = never; DEFAULT_TABLES overrides None (the test-suite pins
        the bucket path that way)."""
        self.lib = lib or B.load()
        self.matrices = matrices
        self.pk = pk
        if pk is None:  # witness-map-only context (R1CSToQAP use)
            if n_vars is None:
                # the two producers of ConstraintMatrices disagree on num_witness_variables
                # (read_zkey: n_vars - n_public, src/zkey.rs:183; R1CS: n_wires - num_inputs), so the
                # witness length cannot be derived from the counts: it is max wire index + 1 at least
                raise G16Error(B.G16_ERR_INVALID, "a witness-map-only Prover needs n_vars (len(full_assignment))")
            need = matrices.num_constraints + matrices.num_instance_variables
            dom = 1
            while dom < need:
                dom <<= 1
            kd = B.KeyDesc()
            kd.n_vars, kd.n_public, kd.domain_size = n_vars, matrices.num_instance_variables - 1, dom
        else:
            kd = pk.to_c()
        self.n_vars = kd.n_vars
        self.domain_size = kd.domain_size
        opt = B.Options()
        opt.device, opt.rank, opt.world = device, rank, world
        opt.window_bits, opt.planes = window_bits, planes
        opt.dist_wm = 1 if dist_wm else 0
        opt.reduction = REDUCTIONS[reduction]
        opt.shard = SHARD_MODES[shard]
        opt.fixed_tables = DEFAULT_TABLES if tables is None else int(tables)
        self.dist_wm = bool(opt.dist_wm)
        self.rank, self.world = rank, world
        a, b = matrices.a.to_c(), matrices.b.to_c()
        ctx = C.c_void_p()
        self.devices = list(devices) if devices is not None else None
        self._donor = sibling_of                          # keeps the lender alive
        if sibling_of is not None:
            st = self.lib.g16_ctx_create_sibling(sibling_of.ctx, C.byref(kd), C.byref(a), C.byref(b),
                                                 matrices.num_constraints, C.byref(opt), C.byref(ctx))
        elif self.devices is not None:
            opt.dist_wm = -1 if dist_wm is None else 0   # None: force a replicated witness map
            ids = (C.c_int * len(self.devices))(*self.devices)
            st = self.lib.g16_ctx_create_multi(C.byref(kd), C.byref(a), C.byref(b), matrices.num_constraints,
                                               ids, len(self.devices), C.byref(opt), C.byref(ctx))
        else:
            st = self.lib.g16_ctx_create(C.byref(kd), C.byref(a), C.byref(b), matrices.num_constraints,
                                         C.byref(opt), C.byref(ctx))
        self.lib.check(st, None)
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.g16_ctx_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    # -- CircomReduction::witness_map_from_matrices
    def witness_map(self, full_assignment) -> np.ndarray:
        w = _as_fr(full_assignment, self.lib)
        h = np.empty((self.domain_size, 4), dtype=np.uint64)
        self.lib.check(self.lib.g16_witness_map(self.ctx, _np_ptr(w), w.shape[0], _np_ptr(h)), self.ctx)
        return h

    def msm_g1(self, which: int, scalars) -> bytes:
        s = _as_fr(scalars, self.lib)
        out = np.empty(64, dtype=np.uint8)
        self.lib.check(self.lib.g16_msm_g1(self.ctx, which, _np_ptr(s), s.shape[0], _np_ptr(out)), self.ctx)
        return out.tobytes()

    def msm_g2(self, scalars) -> bytes:
        s = _as_fr(scalars, self.lib)
        out = np.empty(128, dtype=np.uint8)
        self.lib.check(self.lib.g16_msm_g2(self.ctx, _np_ptr(s), s.shape[0], _np_ptr(out)), self.ctx)
        return out.tobytes()

    def witness_map_dev(self, w_dev_ptr: int, h_dev_ptr: int):
        self.lib.check(self.lib.g16_witness_map_dev(self.ctx, C.c_void_p(w_dev_ptr), self.n_vars,
                                                    C.c_void_p(h_dev_ptr)), self.ctx)

    def msm_g1_dev(self, which: int, scalars_dev_ptr: int, count: int) -> bytes:
        out = np.empty(64, dtype=np.uint8)
        self.lib.check(self.lib.g16_msm_g1_dev(self.ctx, which, C.c_void_p(scalars_dev_ptr), count,
                                               _np_ptr(out)), self.ctx)
        return out.tobytes()

    def msm_g2_dev(self, scalars_dev_ptr: int, count: int) -> bytes:
        out = np.empty(128, dtype=np.uint8)
        self.lib.check(self.lib.g16_msm_g2_dev(self.ctx, C.c_void_p(scalars_dev_ptr), count,
                                               _np_ptr(out)), self.ctx)
        return out.tobytes()

    def prove(self, r, s, full_assignment) -> Proof:
        w = _as_fr(full_assignment, self.lib)
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        out = np.empty(B.G16_PROOF_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]), _np_ptr(w),
                                          w.shape[0], _np_ptr(out)), self.ctx)
        return Proof(out.tobytes())

    def prove_dev(self, r, s, w_dev_ptr: int) -> Proof:
        """witness already resident in HBM (device pointer to n_vars x 32 bytes)"""
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        out = np.empty(B.G16_PROOF_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove_dev(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]),
                                              C.c_void_p(w_dev_ptr), self.n_vars, _np_ptr(out)), self.ctx)
        return Proof(out.tobytes())

    def prove_partial(self, r, s, full_assignment=None, w_dev_ptr: Optional[int] = None) -> bytes:
        """this rank's G16_PARTIAL_BYTES record: its A, B1, B2, L, H sums and s*A, r*B1"""
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        out = np.empty(B.G16_PARTIAL_BYTES, dtype=np.uint8)
        if w_dev_ptr is not None:
            st = self.lib.g16_prove_partial_dev(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]),
                                                C.c_void_p(w_dev_ptr), self.n_vars, _np_ptr(out))
        else:
            w = _as_fr(full_assignment, self.lib)
            st = self.lib.g16_prove_partial(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]), _np_ptr(w),
                                            w.shape[0], _np_ptr(out))
        self.lib.check(st, self.ctx)
        return out.tobytes()

    # -- fully sharded prover (dist_wm=True): three phases around two all-to-all exchanges
    def exchange_bytes(self) -> int:
        return int(self.lib.g16_dist_exchange_bytes(self.ctx))

    def dist_phase1(self, r, s, w_dev_ptr: int, send_ptr: int):
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        self.lib.check(self.lib.g16_prove_dist_phase1(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]),
                                                      C.c_void_p(w_dev_ptr), self.n_vars,
                                                      C.c_void_p(send_ptr)), self.ctx)

    def dist_phase2(self, recv_ptr: int, send_ptr: int):
        self.lib.check(self.lib.g16_prove_dist_phase2(self.ctx, C.c_void_p(recv_ptr),
                                                      C.c_void_p(send_ptr)), self.ctx)

    def dist_phase3(self, recv_ptr: int) -> bytes:
        out = np.empty(B.G16_PARTIAL_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove_dist_phase3(self.ctx, C.c_void_p(recv_ptr), _np_ptr(out)),
                       self.ctx)
        return out.tobytes()

    def prove_finish(self, r, s, partials: bytes) -> Proof:
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        world = len(partials) // B.G16_PARTIAL_BYTES
        buf = np.frombuffer(partials, dtype=np.uint8)
        out = np.empty(B.G16_PROOF_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove_finish(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]),
                                                 _np_ptr(buf), world, _np_ptr(out)), self.ctx)
        return Proof(out.tobytes())

    def witness_buffer(self) -> int:
        return int(self.lib.g16_witness_buffer(self.ctx) or 0)

    def upload_witness(self, full_assignment) -> int:
        """witness resident in HBM (on every device of a multi-device ctx); returns the pointer to
        pass to prove_dev()"""
        w = _as_fr(full_assignment, self.lib)
        self.lib.check(self.lib.g16_witness_upload(self.ctx, _np_ptr(w), w.shape[0]), self.ctx)
        return self.witness_buffer()

    def witness_host_buffer(self) -> np.ndarray:
        """(n_vars, 4) uint64 view of the ctx's page-locked staging buffer (g16_witness_host_buffer)"""
        p = self.lib.g16_witness_host_buffer(self.ctx)
        if not p:
            raise G16Error(B.G16_ERR_HIP, "pinned allocation failed")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(self.n_vars * 4,))
        return arr.reshape(self.n_vars, 4)

    # -- device-side hand-offs for a host framework that owns a stream (torch + RCCL)
    def set_exchange_stream(self, hip_stream: int, enabled=True):
        self.lib.check(self.lib.g16_dist_set_exchange_stream(self.ctx, C.c_void_p(hip_stream),
                                                             1 if enabled else 0), self.ctx)

    def partial_buffer(self) -> int:
        return int(self.lib.g16_partial_buffer(self.ctx) or 0)

    def gather_buffer(self) -> int:
        return int(self.lib.g16_gather_buffer(self.ctx) or 0)

    def dist_phase3_dev(self, recv_ptr: int):
        """phase 3 with the record left in partial_buffer() (needs set_exchange_stream)"""
        self.lib.check(self.lib.g16_prove_dist_phase3(self.ctx, C.c_void_p(recv_ptr), None), self.ctx)

    def prove_finish_dev(self, r, s) -> Proof:
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        out = np.empty(B.G16_PROOF_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove_finish_dev(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]),
                                                     _np_ptr(out)), self.ctx)
        return Proof(out.tobytes())

    def attach_rccl(self, nccl_comm: int):
        """hand the library an ncclComm_t the host created over the same ranks (g16_dist_attach_rccl): the
        collectives of prove_dist() are then issued by the library itself"""
        self.lib.check(self.lib.g16_dist_attach_rccl(self.ctx, C.c_void_p(nccl_comm)), self.ctx)

    def rccl_ranks(self) -> int:
        return int(self.lib.g16_dist_rccl_ranks(self.ctx))

    def prove_dist(self, r, s, w_dev_ptr: int) -> Proof:
        """one whole sharded proof of this rank through the attached communicator (g16_prove_dist)"""
        rs = _as_fr([r, s], self.lib) if not isinstance(r, np.ndarray) else np.stack([r, s])
        out = np.empty(B.G16_PROOF_BYTES, dtype=np.uint8)
        self.lib.check(self.lib.g16_prove_dist(self.ctx, _np_ptr(rs[0:1]), _np_ptr(rs[1:2]), C.c_void_p(w_dev_ptr),
                                               self.n_vars, _np_ptr(out)), self.ctx)
        return Proof(out.tobytes())

    def set_profiling(self, on: bool):
        self.lib.check(self.lib.g16_set_profiling(self.ctx, 1 if on else 0), self.ctx)

    def stage_times(self):
        ms = (C.c_float * B.G16_N_STAGES)()
        cnt = (C.c_uint32 * B.G16_N_STAGES)()
        self.lib.check(self.lib.g16_stage_times(self.ctx, ms, cnt), self.ctx)
        return {self.lib.g16_stage_name(i).decode(): (float(ms[i]), int(cnt[i]))
                for i in range(B.G16_N_STAGES)}

    def links(self):
        """multi-device ctx: the create-time link probe (g16_multi_links) as
        {"probe_bytes", "gbps": [[src][dst]], "echo_us": [[src][dst]]}"""
        n = self.info()["devices"]
        gb, us = (C.c_float * (n * n))(), (C.c_float * (n * n))()
        pb = C.c_uint64(0)
        self.lib.check(self.lib.g16_multi_links(self.ctx, gb, us, n * n, C.byref(pb)), self.ctx)
        return {"probe_bytes": int(pb.value), "gbps": [[float(gb[a * n + b]) for b in range(n)] for a in range(n)],
                "echo_us": [[float(us[a * n + b]) for b in range(n)] for a in range(n)]}

    def info(self):
        out = (C.c_uint32 * 16)()
        self.lib.check(self.lib.g16_ctx_info(self.ctx, out), self.ctx)
        keys = ["c_w", "W_w", "planes_w", "D_w", "c_h", "W_h", "planes_h", "D_h", "domain_size",
                "log_n", "shard_w", "shard_h", "devices", "shard_mode", "peer_access", "fixed_tables"]
        d = dict(zip(keys, list(out)))
        d["shard_mode"] = {0: "none", 1: "points", 2: "buckets"}.get(d["shard_mode"], "?")
        d["sparse_b"] = (d["fixed_tables"] >> 1) & 1       # out[15]: bit 0 = fixed-base tables, bit 1 = filtered B view
        d["fixed_tables"] &= 1
        return d


def _transpose_csr(m: Csr, n_cols: int, extra=None) -> Csr:
    """CSR (rows = constraints) -> CSR of the transpose (rows = wires).  extra: optional
    (rows, cols, coeff) triplets appended before transposing."""
    counts = np.diff(m.row_ptr.astype(np.int64))
    rows = np.repeat(np.arange(m.num_rows, dtype=np.int64), counts)
    cols = m.col.astype(np.int64)
    vals = m.coeff
    if extra is not None:
        rows = np.concatenate([rows, np.asarray(extra[0], dtype=np.int64)])
        cols = np.concatenate([cols, np.asarray(extra[1], dtype=np.int64)])
        vals = np.concatenate([vals, np.asarray(extra[2], dtype=np.uint64).reshape(-1, 4)])
    order = np.argsort(cols, kind="stable")
    rp = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n_cols), out=rp[1:])
    return Csr(rp.astype(np.uint32), rows[order].astype(np.uint32), vals[order])


REDUCTIONS = {"circom": 0, "libsnark": 1}
SHARD_MODES = {"auto": B.SHARD_AUTO, "points": B.SHARD_POINTS, "buckets": B.SHARD_BUCKETS}


def trapdoor_setup(a: Csr, b: Csr, c: Csr, n_vars: int, n_public: int, toxic: Sequence[int],
                   device=0, lib: Optional[B.Library] = None, reduction: str = "circom") -> ProvingKey:
    """Known-toxic-waste setup on the GPU (g16_setup_create_ex): the key
    Groth16::generate_random_parameters_with_reduction::<QAP> would produce for
    (tau, alpha, beta, gamma, delta) = toxic, QAP = CircomReduction ("circom", snarkjs-compatible)
    or LibsnarkReduction ("libsnark", arkworks' default: reference tests/groth16.rs:25).
    Used to mint the synthetic BASELINE keys."""
    lib = lib or B.load()
    m = a.num_rows
    ni = n_public + 1
    one = fr_from_ints([1], lib)
    at = _transpose_csr(a, n_vars, (np.arange(m, m + ni), np.arange(ni), np.tile(one, (ni, 1))))
    bt = _transpose_csr(b, n_vars)
    ct = _transpose_csr(c, n_vars)
    tox = fr_from_ints(list(toxic), lib)
    h = C.c_void_p()
    cat, cbt, cct = at.to_c(), bt.to_c(), ct.to_c()
    st = lib.g16_setup_create_ex(device, C.byref(cat), C.byref(cbt), C.byref(cct), n_vars, n_public, m,
                                 _np_ptr(tox), REDUCTIONS[reduction], C.byref(h))
    if st != B.G16_OK:
        raise (SynthesisError if st == B.G16_ERR_DOMAIN_TOO_LARGE else G16Error)(st, "g16_setup_create failed")
    handle = _Handle(lib, h, lib.g16_setup_destroy)
    kd = B.KeyDesc()
    icp = C.c_void_p()
    cnt = C.c_uint32()
    gamma = (C.c_uint8 * 128)()
    lib.check(lib.g16_setup_key(h, C.byref(kd), C.byref(icp), C.byref(cnt), gamma))

    def view(ptr, count, width):
        if count == 0:
            return np.zeros((0, width), dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * width,)).reshape(count, width)

    vk = VerifyingKey(bytes(kd.alpha_g1), bytes(kd.beta_g2), bytes(gamma), bytes(kd.delta_g2),
                      view(icp, cnt.value, 64).copy())
    return ProvingKey(kd.n_vars, kd.n_public, kd.domain_size, vk, bytes(kd.beta_g1), bytes(kd.delta_g1),
                      view(kd.a_query, n_vars, 64), view(kd.b_g1_query, n_vars, 64),
                      view(kd.b_g2_query, n_vars, 128), view(kd.l_query, n_vars - n_public - 1, 64),
                      view(kd.h_query, kd.domain_size, 64), keepalive=handle)


def verify_batch(vk: "VerifyingKey", proofs, public_inputs, device=0, lib: Optional[B.Library] = None):
    """Groth16::process_vk + verify_with_processed_vk (reference src/zkey.rs:868-870,914-916) for a
    batch under one key on the GPU (g16_verify_batch).  proofs: Proof objects or 256-byte strings;
    public_inputs: one sequence of n_public values (ints or Montgomery rows) per proof.  Returns a
    list of bools."""
    lib = lib or B.load()
    raw = b"".join(p.raw if isinstance(p, Proof) else bytes(p) for p in proofs)
    n = len(raw) // B.G16_PROOF_BYTES
    ic = np.ascontiguousarray(vk.gamma_abc_g1, dtype=np.uint8).reshape(-1, 64)
    n_pub = ic.shape[0] - 1
    if len(public_inputs) != n:
        raise G16Error(B.G16_ERR_INVALID, "one public-input vector per proof")
    flat = []
    for pi in public_inputs:
        if len(pi) != n_pub:
            raise G16Error(B.G16_ERR_INVALID, "MalformedVerifyingKey: wrong number of public inputs")
        flat.extend(pi)
    pubs = _as_fr(flat, lib) if (flat and not isinstance(flat[0], np.ndarray)) else \
        (np.ascontiguousarray(np.stack(flat)) if flat else np.zeros((0, 4), np.uint64))
    d = B.VkDesc()
    C.memmove(d.alpha_g1, bytes(vk.alpha_g1), 64)
    C.memmove(d.beta_g2, bytes(vk.beta_g2), 128)
    C.memmove(d.gamma_g2, bytes(vk.gamma_g2), 128)
    C.memmove(d.delta_g2, bytes(vk.delta_g2), 128)
    d.ic, d.ic_count = ic.ctypes.data, ic.shape[0]
    buf = np.frombuffer(raw, dtype=np.uint8)
    ok = np.zeros(max(n, 1), dtype=np.uint8)
    st = lib.g16_verify_batch(device, C.byref(d), _np_ptr(buf), _np_ptr(pubs), n, _np_ptr(ok))
    if st != B.G16_OK:
        raise G16Error(st, "g16_verify_batch failed")
    return [bool(x) for x in ok[:n]]


class _Reduction:
    """R1CSToQAP::witness_map_from_matrices on the GPU; the subclass names the QAP."""
    NAME = "circom"

    @classmethod
    def witness_map_from_matrices(cls, matrices: ConstraintMatrices, num_inputs: int,
                                  num_constraints: int, full_assignment, lib=None, device=0):
        if num_inputs != matrices.num_instance_variables or num_constraints != matrices.num_constraints:
            raise G16Error(B.G16_ERR_INVALID, "num_inputs/num_constraints do not match the matrices")
        attr = "_wm_prover_" + cls.NAME
        pr = getattr(matrices, attr, None)
        if pr is None:
            n_vars = len(full_assignment)
            pr = Prover(None, matrices, device=device, lib=lib, n_vars=n_vars, reduction=cls.NAME)
            setattr(matrices, attr, pr)
        return pr.witness_map(full_assignment)


class CircomReduction(_Reduction):
    """R1CSToQAP impl used for circom/snarkjs keys (reference src/circom/qap.rs:12-106)."""
    NAME = "circom"


class LibsnarkReduction(_Reduction):
    """ark_groth16::LibsnarkReduction, the default QAP of `Groth16<Bn254>` (arkworks-generated keys,
    reference tests/groth16.rs:9,25-35).  Returns the n coefficients of h."""
    NAME = "libsnark"


class Groth16:
    """Groth16::<Bn254, CircomReduction> entry points of the proving path."""

    @staticmethod
    def _prover(pk: ProvingKey, matrices: ConstraintMatrices, **kw) -> Prover:
        kw.setdefault("reduction", getattr(pk, "reduction", "circom"))
        if pk._prover is None or pk._prover.matrices is not matrices:
            pk._prover = Prover(pk, matrices, **kw)
        return pk._prover

    @staticmethod
    def generate_random_parameters_with_reduction(r1cs: "R1CS", rng=None, reduction: str = "libsnark",
                                                  device=0, lib=None) -> ProvingKey:
        """Groth16::<Bn254, QAP>::generate_random_parameters_with_reduction(circuit, rng) (reference
        tests/groth16.rs:25, QAP defaulting to LibsnarkReduction there): toxic waste from rng, key
        minted on the GPU.  The key remembers its reduction; Groth16.prove uses it."""
        rng = rng or random.SystemRandom()
        toxic = [rng.randrange(1, FR_MODULUS) for _ in range(5)]
        pk = trapdoor_setup(r1cs.a, r1cs.b, r1cs.c, r1cs.num_variables, r1cs.num_inputs - 1, toxic,
                            device=device, lib=lib, reduction=reduction)
        pk.reduction = reduction
        return pk

    @staticmethod
    def verify(vk: "VerifyingKey", public_inputs, proof, **kw) -> bool:
        """Groth16::verify_with_processed_vk(&process_vk(&vk), inputs, &proof) on the GPU"""
        return verify_batch(vk, [proof], [list(public_inputs)], **kw)[0]

    @staticmethod
    def create_proof_with_reduction_and_matrices(pk: ProvingKey, r, s, matrices: ConstraintMatrices,
                                                 num_inputs: int, num_constraints: int,
                                                 full_assignment, **kw) -> Proof:
        """Argument order of reference benches/groth16.rs:52-60 / src/zkey.rs:903-911."""
        if num_inputs != matrices.num_instance_variables or num_constraints != matrices.num_constraints:
            raise G16Error(B.G16_ERR_INVALID, "num_inputs/num_constraints do not match the matrices")
        return Groth16._prover(pk, matrices, **kw).prove(r, s, full_assignment)

    @staticmethod
    def prove(pk: ProvingKey, matrices: ConstraintMatrices, circuit: CircomCircuit, rng=None, **kw) -> Proof:
        """SNARK::prove(&pk, circuit, rng) (reference src/zkey.rs:866): r, s <- rng, then the matrices
        entry (rebuilding a ConstraintSystem per proof is the serial host work this path avoids)."""
        rng = rng or random.SystemRandom()
        r = rng.randrange(FR_MODULUS)
        s = rng.randrange(FR_MODULUS)
        # the assignment generate_constraints would allocate (circuit.rs:35-58): through the wire
        # mapping when the circuit carries one, so that it matches get_public_inputs()
        w = circuit.full_assignment()
        return Groth16.create_proof_with_reduction_and_matrices(
            pk, r, s, matrices, matrices.num_instance_variables, matrices.num_constraints, w, **kw)
