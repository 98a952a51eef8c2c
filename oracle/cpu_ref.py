"""oracle/cpu_ref.py -- TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/libg16_cpu_oracle.so
(the multithreaded C restatement of the reference's CPU proving path, oracle/groth16_cpu.c).
Used by tests/ as the large-size checker and by bench.py's cpu_baseline leg; never by the product."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Csr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("coeff", C.c_void_p), ("nnz", C.c_uint64)]


class _Key(C.Structure):
    _fields_ = [("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32),
                ("a_query", C.c_void_p), ("b_g1_query", C.c_void_p), ("b_g2_query", C.c_void_p),
                ("l_query", C.c_void_p), ("h_query", C.c_void_p),
                ("alpha_g1", C.c_uint8 * 64), ("beta_g1", C.c_uint8 * 64), ("delta_g1", C.c_uint8 * 64),
                ("beta_g2", C.c_uint8 * 128), ("delta_g2", C.c_uint8 * 128)]


VARIANT = None  # "adx" (built with -mbmi2 -madx) or "baseline": which build lib() loaded


def _cpu_has_adx_bmi2():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                return "adx" in flags and "bmi2" in flags
    except OSError:
        pass
    return False


def host_cpu_grant(root="/sys/fs/cgroup", aff=None):
    """Cores this process can really use: the affinity mask cut by the cgroup CPU quota (cgroup v2 cpu.max, v1
    cfs_quota_us / cfs_period_us).  A gpurun box shows 256 logical CPUs and grants 16 (cpu.max = 1600000 100000,
    `scripts/host_cores_probe.py`): threads beyond the quota are throttled, not run.  -> (cores, description)"""
    if aff is None:
        try:
            aff = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            aff = os.cpu_count() or 1
    quota, how = None, "no cgroup CPU quota"
    try:
        q, per = open(os.path.join(root, "cpu.max")).read().split()[:2]
        if q != "max":
            quota, how = float(q) / float(per), f"cgroup v2 cpu.max = {q} {per}"
    except (OSError, ValueError):
        try:
            q = int(open(os.path.join(root, "cpu", "cpu.cfs_quota_us")).read())
            per = int(open(os.path.join(root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0:
                quota, how = q / per, f"cgroup v1 cfs quota {q} / {per}"
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return cores, f"{aff} CPUs in the affinity mask, {how}"


_OMP_DEFAULT = None


def omp_default_threads():
    """OpenMP's own thread count before lib() cut it to the host's grant"""
    lib()
    return _OMP_DEFAULT


def lib():
    global _LIB, VARIANT, _OMP_DEFAULT
    if _LIB is None:
        VARIANT = "adx" if _cpu_has_adx_bmi2() and not os.environ.get("G16_CPU_BASELINE_ISA") else "baseline"
        name = "libg16_cpu_oracle_adx.so" if VARIANT == "adx" else "libg16_cpu_oracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE])
        _LIB = C.CDLL(path)
        _LIB.g16cpu_max_threads.restype = C.c_int
        # threads far beyond the cgroup CPU quota are throttled, not run: size the pool to TWICE what the host
        # grants (not 1x: the window-parallel MSM has 15-19 tasks, and 17 tasks on 16 threads take two rounds --
        # profiles/r06_cpu_threads_sweep.txt: 2^20 proof 5.1 s on 16 threads, 3.0-3.2 s on 24-64, 3.7 s on 128)
        _OMP_DEFAULT = _LIB.g16cpu_max_threads()
        want = int(os.environ.get("G16_CPU_THREADS", "0")) or min(_OMP_DEFAULT, 2 * host_cpu_grant()[0])
        if want != _OMP_DEFAULT:
            _LIB.g16cpu_set_threads(want)
    return _LIB


def variant():
    lib()
    return VARIANT


def set_threads(n):
    lib().g16cpu_set_threads(int(n))


def set_msm_chunks(n):
    """tasks per MSM window: 1 = one task per window (ark-ec's msm_bigint under rayon); n > 1 cuts every
    window's bases into n chunks so that windows x n tasks keep all host threads busy (same result)"""
    lib().g16cpu_set_msm_chunks(int(n))


def set_msm_window(c):
    """window bits of every MSM (0 = ark-ec's ln-based choice)"""
    lib().g16cpu_set_msm_window(int(c))


def max_threads():
    return lib().g16cpu_max_threads()


def _csr(rp, col, coeff):
    c = _Csr()
    c.row_ptr, c.col, c.coeff, c.nnz = rp.ctypes.data, col.ctypes.data, coeff.ctypes.data, col.shape[0]
    return c


def witness_map(a, b, num_inputs, m, w, reduction="circom"):
    """a, b: objects with row_ptr/col/coeff numpy arrays (Montgomery); w (N,4) uint64 Montgomery"""
    need = m + num_inputs
    n = 1
    while n < need:
        n <<= 1
    h = np.empty((n, 4), dtype=np.uint64)
    dom = C.c_uint32()
    ca, cb = _csr(a.row_ptr, a.col, a.coeff), _csr(b.row_ptr, b.col, b.coeff)
    fn = lib().g16cpu_witness_map_libsnark if reduction == "libsnark" else lib().g16cpu_witness_map
    st = fn(C.byref(ca), C.byref(cb), C.c_uint32(num_inputs), C.c_uint32(m),
            C.c_void_p(w.ctypes.data), C.c_void_p(h.ctypes.data), C.byref(dom))
    if st == 2:
        raise ValueError("PolynomialDegreeTooLarge")
    assert st == 0 and dom.value == n
    return h


def msm_g1(bases, scalars_mont):
    out = np.empty(64, dtype=np.uint8)
    lib().g16cpu_msm_g1(C.c_void_p(bases.ctypes.data), C.c_void_p(scalars_mont.ctypes.data),
                        C.c_size_t(scalars_mont.shape[0]), C.c_void_p(out.ctypes.data))
    return out.tobytes()


def msm_g2(bases, scalars_mont):
    out = np.empty(128, dtype=np.uint8)
    lib().g16cpu_msm_g2(C.c_void_p(bases.ctypes.data), C.c_void_p(scalars_mont.ctypes.data),
                        C.c_size_t(scalars_mont.shape[0]), C.c_void_p(out.ctypes.data))
    return out.tobytes()


def prove(pk, mats, r_mont, s_mont, w, want_h=False, reduction="circom"):
    """pk: circom_compat_amd.ProvingKey-like (packed numpy arrays); mats: ConstraintMatrices-like"""
    k = _Key()
    k.n_vars, k.n_public, k.domain_size = pk.n_vars, pk.n_public, pk.domain_size
    for name in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        setattr(k, name, getattr(pk, name).ctypes.data)
    C.memmove(k.alpha_g1, bytes(pk.vk.alpha_g1), 64)
    C.memmove(k.beta_g1, bytes(pk.beta_g1), 64)
    C.memmove(k.delta_g1, bytes(pk.delta_g1), 64)
    C.memmove(k.beta_g2, bytes(pk.vk.beta_g2), 128)
    C.memmove(k.delta_g2, bytes(pk.vk.delta_g2), 128)
    ca = _csr(mats.a.row_ptr, mats.a.col, mats.a.coeff)
    cb = _csr(mats.b.row_ptr, mats.b.col, mats.b.coeff)
    out = np.empty(256, dtype=np.uint8)
    h = np.empty((pk.domain_size, 4), dtype=np.uint64) if want_h else None
    st = lib().g16cpu_prove_ex(C.byref(k), C.byref(ca), C.byref(cb), C.c_uint32(mats.num_constraints),
                               C.c_void_p(r_mont.ctypes.data), C.c_void_p(s_mont.ctypes.data),
                               C.c_void_p(w.ctypes.data), C.c_void_p(out.ctypes.data),
                               C.c_void_p(h.ctypes.data) if want_h else None,
                               C.c_int(1 if reduction == "libsnark" else 0))
    if st == 2:
        raise ValueError("PolynomialDegreeTooLarge")
    assert st == 0, st
    return (out.tobytes(), h) if want_h else out.tobytes()


def _canon_arr(ks):
    a = np.zeros((len(ks), 4), dtype=np.uint64)
    for i, k in enumerate(ks):
        for j in range(4):
            a[i, j] = (k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return a


def g1_mul_batch(point_bytes, ks):
    """[k * P for k in ks] as 64-byte packed affine points (P: 64 bytes, ks: ints < r)"""
    a = _canon_arr(ks)
    out = np.empty((len(ks), 64), dtype=np.uint8)
    p = np.frombuffer(bytes(point_bytes), dtype=np.uint8).copy()
    lib().g16cpu_g1_mul_batch(C.c_void_p(p.ctypes.data), C.c_void_p(a.ctypes.data), C.c_size_t(len(ks)),
                              C.c_void_p(out.ctypes.data))
    return out


def g2_mul_batch(point_bytes, ks):
    a = _canon_arr(ks)
    out = np.empty((len(ks), 128), dtype=np.uint8)
    p = np.frombuffer(bytes(point_bytes), dtype=np.uint8).copy()
    lib().g16cpu_g2_mul_batch(C.c_void_p(p.ctypes.data), C.c_void_p(a.ctypes.data), C.c_size_t(len(ks)),
                              C.c_void_p(out.ctypes.data))
    return out


def fft(data, log_n, inverse=False):
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    lib().g16cpu_fft(C.c_void_p(a.ctypes.data), C.c_int(log_n), C.c_int(1 if inverse else 0))
    return a
