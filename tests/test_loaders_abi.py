"""CPU-only tests of the product library's host side: the C ABI exports every symbol include/*.h
declares, the loaders reproduce the reference's golden bytes / values / error behaviour, and the
product path fails loudly without a GPU (no fallback)."""
import ctypes as C
import json
import os
import sys
import re

import numpy as np
import pytest

import bn254_ref as o
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec(golden):
    return json.load(open(os.path.join(golden, "reference_vectors.json")))


def test_library_exports_every_declared_symbol(gpulib):
    declared = set()
    for hdr in ("g16_amd.h", "g16_loaders.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"#ifdef G16_DEBUG_ABI.*?#endif", "", src, flags=re.S)   # measurement builds only: not in the product
        declared |= set(re.findall(r"\b(g16_[a-z0-9_]+)\s*\(", src))
    assert declared, "no declarations parsed"
    from circom_compat_amd import _binding
    assert declared == set(_binding.ABI_SYMBOLS), declared ^ set(_binding.ABI_SYMBOLS)
    assert gpulib.missing == []
    for name in declared:
        assert hasattr(gpulib.L, name), name
    # VERDICT r5 hygiene: nothing named g16_debug_* leaves the product library
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", gpulib.path], capture_output=True, text=True).stdout
    assert "g16_debug_" not in syms


def test_no_cpu_fallback(gpulib, golden):
    """without a HIP device the product refuses to run (G16_ERR_NO_DEVICE), it never falls back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib=gpulib)
    with pytest.raises(cc.G16Error) as e:
        cc.Prover(pk, mats, lib=gpulib)
    assert e.value.status == 4 and "no CPU fallback" in str(e.value)
    arr = H.fr_mont_arr([1, 2, 3, 4])
    assert gpulib.g16_fft_in_place(0, arr.ctypes.data, 2, 0, 0) == 4


def test_package_does_not_touch_oracle_or_emulator():
    """the product package never imports/links the checker or the emulator"""
    pkg = os.path.join(ROOT, "circom_compat_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                s = open(os.path.join(dirpath, f)).read()
                assert "bn254_ref" not in s and "cpu_ref" not in s and "groth16_cpu" not in s, f
                if f != "Makefile" and f != "common.h":
                    assert "libg16_emu" not in s, f


def test_read_zkey_golden(gpulib, vec, golden):
    """reference src/zkey.rs:519-543,545-779 through the C++ loader"""
    import circom_compat_amd as cc
    pk, mats = cc.read_zkey(os.path.join(golden, "test.zkey"), lib=gpulib)
    hd = vec["test_zkey_header"]
    assert (pk.n_vars, pk.n_public, pk.domain_size) == (hd["n_vars"], hd["n_public"], hd["domain_size"])
    tz = vec["test_zkey"]
    for name, arr in (("ic", pk.vk.gamma_abc_g1), ("a_query", pk.a_query), ("b_g1_query", pk.b_g1_query),
                      ("b_g2_query", pk.b_g2_query), ("l_query", pk.l_query), ("h_query", pk.h_query)):
        assert arr.tolist() == tz[name], name
    opk, om = o.read_zkey(open(os.path.join(golden, "test.zkey"), "rb").read())
    assert bytes(pk.vk.alpha_g1) == o.g1_to_bytes(opk["alpha_g1"]) and bytes(pk.beta_g1) == o.g1_to_bytes(opk["beta_g1"])
    assert bytes(pk.vk.beta_g2) == o.g2_to_bytes(opk["beta_g2"]) and bytes(pk.vk.gamma_g2) == o.g2_to_bytes(opk["gamma_g2"])
    assert bytes(pk.delta_g1) == o.g1_to_bytes(opk["delta_g1"]) and bytes(pk.vk.delta_g2) == o.g2_to_bytes(opk["delta_g2"])
    # matrices(): src/zkey.rs:151-196
    assert (mats.num_instance_variables, mats.num_witness_variables, mats.num_constraints) == (2, 3, 1)
    assert (mats.a_num_non_zero, mats.b_num_non_zero, mats.c_num_non_zero) == (1, 1, 0)
    assert mats.a.col.tolist() == [2] and mats.b.col.tolist() == [3]
    assert H.fr_from_mont_arr(mats.a.coeff) == [o.R_MOD - 1] and H.fr_from_mont_arr(mats.b.coeff) == [1]
    # from memory too, and a truncated file is an error, not a crash
    data = open(os.path.join(golden, "test.zkey"), "rb").read()
    pk2, _ = cc.read_zkey(data, lib=gpulib)
    assert np.array_equal(pk2.h_query, pk.h_query)
    with pytest.raises(cc.SerializationError):
        cc.read_zkey(data[:1500], lib=gpulib)


def test_zkey_writer_roundtrip_through_loader(gpulib):
    """oracle's snarkjs-format writer -> product loader: exercises Coefs(4) incl. the dropped
    public-input rows (src/zkey.rs:171-175) on a multi-row circuit"""
    import circom_compat_amd as cc
    cons, w, n_vars, n_pub = H.squaring_chain(3)
    opk = o.trapdoor_setup(cons, n_vars, n_pub, 3, 5, 7, 11, 13)
    blob = o.write_zkey(opk, o.coefs_from_r1cs(cons, n_pub))
    pk, mats = cc.read_zkey(blob, lib=gpulib)
    rpk, rm = o.read_zkey(blob)
    assert rm["num_constraints"] == len(cons) == mats.num_constraints
    a_rows, b_rows = o.matrices_from_r1cs(cons)
    assert rm["a"] == a_rows and rm["b"] == b_rows
    want = H.pk_from_oracle(opk)
    for name in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        assert np.array_equal(getattr(pk, name), getattr(want, name)), name
    rows = lambda m: [[(cf, int(c)) for cf, c in zip(H.fr_from_mont_arr(m.coeff[m.row_ptr[i]:m.row_ptr[i + 1]]),
                                                     m.col[m.row_ptr[i]:m.row_ptr[i + 1]])] for i in range(m.num_rows)]
    assert rows(mats.a) == a_rows and rows(mats.b) == b_rows


def test_r1cs_loader_golden_and_errors(gpulib, vec, golden):
    """reference src/circom/r1cs_reader.rs:257-338 + error paths :57-69,163-189,232-247"""
    import circom_compat_amd as cc
    good = bytes.fromhex(vec["r1cs_sample_hex"])
    f = cc.R1CSFile(good, lib=gpulib)
    e = vec["r1cs_sample_expect"]
    h = f.header
    assert (f.version, h.field_size, h.n_wires, h.n_pub_out, h.n_pub_in, h.n_prv_in, h.n_labels, h.n_constraints) == \
        (e["version"], e["field_size"], e["n_wires"], e["n_pub_out"], e["n_pub_in"], e["n_prv_in"], e["n_labels"], e["n_constraints"])
    assert bytes(h.prime) == o.R1CS_PRIME_LE
    assert f.a.row_ptr[1] - f.a.row_ptr[0] == e["c0_a_len"]
    assert int(f.a.col[0]) == e["c0_a0"][0] and H.fr_from_mont_arr(f.a.coeff[0:1]) == [e["c0_a0"][1]]
    b2 = f.b.row_ptr[2]
    assert int(f.b.col[b2]) == e["c2_b0"][0] and H.fr_from_mont_arr(f.b.coeff[b2:b2 + 1]) == [e["c2_b0"][1]]
    assert f.c.row_ptr[2] - f.c.row_ptr[1] == e["c1_c_len"]
    assert len(f.wire_mapping) == e["wire_mapping_len"] and f.wire_mapping[1] == e["wire_mapping_1"]
    r = cc.R1CS(f)
    assert (r.num_inputs, r.num_variables, r.num_aux) == (4, 7, 3)
    for mutate, msg in ((lambda d: d.__setitem__(0, 0x73), "Invalid magic number"),
                        (lambda d: d.__setitem__(4, 2), "Unsupported version"),
                        (lambda d: d.__setitem__(24, 31), "This parser only supports 32-byte fields"),
                        (lambda d: (d.__setitem__(16, 0x41), d.insert(88, 0)), "Invalid header section size"),
                        (lambda d: d.__setitem__(28, 2), "This parser only supports bn256"),
                        (lambda d: d.__setitem__(len(d) - 56, 1), "Wire 0 should always be mapped to 0")):
        d = bytearray(good)
        mutate(d)
        with pytest.raises(cc.SerializationError, match=msg):
            cc.R1CSFile(bytes(d), lib=gpulib)
    # fixtures: same values as the oracle parser
    # complex-circuit-10000-10000.r1cs: the reference bench's default circuit (benches/groth16.rs:
    # 87-108); its HEADER section comes first in the file, the other fixtures have the constraints
    # first (r1cs_reader.rs:80-87 accepts any order)
    for name in ("mycircuit.r1cs", "circuit2.r1cs", "complex-circuit-10000-10000.r1cs"):
        data = open(os.path.join(golden, name), "rb").read()
        f = cc.R1CSFile(data, lib=gpulib)
        ref = o.read_r1cs(data)
        for k, m in enumerate((f.a, f.b, f.c)):
            flat = [(int(c), v) for c, v in zip(m.col, H.fr_from_mont_arr(m.coeff))]
            want = [(wdx, cf) for con in ref["constraints"] for wdx, cf in con[k]]
            assert flat == want, (name, k)
        assert [int(x) for x in f.wire_mapping] == ref["wire_mapping"]


def test_reference_bench_circuit_witness_by_forward_solve(gpulib, golden):
    """complex-circuit-10000-10000.r1cs + input a = 3 (test-vectors/complex-circuit/input.json): the
    witness bench.py derives without the WASM calculator satisfies every row of the reference's
    own constraint file (checked with plain integers), and has the template's shape
    (b[0] = a^2, b[i] = b[i-1]^2, 10000 rows, 2 instance variables)."""
    import circom_compat_amd as cc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert json.load(open(os.path.join(golden, "complex-circuit-input.json"))) == {"a": "3"}
    mats, (a, b, c), w, n_vars = bench.complex_circuit(cc)
    assert (mats.num_constraints, mats.num_instance_variables, n_vars) == (10000, 2, 10002)
    ref = o.read_r1cs(open(os.path.join(golden, "complex-circuit-10000-10000.r1cs"), "rb").read())
    lc = lambda terms: sum(cf * w[j] for j, cf in terms) % o.R_MOD
    assert all(lc(A) * lc(B) % o.R_MOD == lc(Cc) for A, B, Cc in ref["constraints"])
    assert w[0] == 1 and w[2] == 3 and w[3] == 9 and w[4] == 81
    assert w[1] == pow(3, 1 << 10000, o.R_MOD)              # c = a^(2^NUM_VARIABLES)


def test_wtns_and_public_inputs(gpulib, golden):
    import circom_compat_amd as cc
    w = cc.read_wtns(os.path.join(golden, "circuit2.wtns"), lib=gpulib)
    want = o.read_wtns(open(os.path.join(golden, "circuit2.wtns"), "rb").read())
    assert cc.fr_to_ints(w, gpulib) == want and H.fr_from_mont_arr(w) == want
    safe = [int(x) for x in json.load(open(os.path.join(golden, "safe-circuit-witness.json")))]
    assert safe == want                                          # witness_calculator.rs:325-360 golden
    r1 = cc.R1CS.from_file(os.path.join(golden, "circuit2.r1cs"), lib=gpulib)
    r1.wire_mapping = None                                       # builder.rs:82-83 disables it
    circ = cc.CircomCircuit(r1, want)
    assert circ.get_public_inputs() == o.get_public_inputs(want, r1.num_inputs)
    assert cc.CircomCircuit(r1, None).get_public_inputs() is None
    vals = [0, 1, o.R_MOD - 1, 12345678901234567890]
    assert cc.fr_to_ints(cc.fr_from_ints(vals, gpulib), gpulib) == vals
    assert np.array_equal(cc.fr_from_ints(vals, gpulib), H.fr_mont_arr(vals))


def test_zkey_copy_mode_survives_a_file_rewritten_under_the_handle(gpulib, golden, tmp_path):
    """g16_zkey_open maps the file (zero-copy views; the file must stay unchanged while the handle is
    open: include/g16_loaders.h).  G16_ZKEY_COPY=1 reads it into owned memory instead -- the key parsed
    from a copy is still intact after the file has been truncated and rewritten."""
    import shutil
    import subprocess
    src = os.path.join(golden, "test.zkey")
    path = str(tmp_path / "copy.zkey")
    shutil.copy(src, path)
    code = ("import os, sys, numpy as np\n"
            "import circom_compat_amd as cc\n"
            "from circom_compat_amd import _binding\n"
            f"lib = _binding.Library({gpulib.path!r})\n"
            f"pk, mats = cc.read_zkey({path!r}, lib=lib)\n"
            "before = pk.h_query.tobytes() + pk.a_query.tobytes()\n"
            f"open({path!r}, 'wb').write(b'\\0' * 64)\n"             # truncate + rewrite under the open handle
            "after = pk.h_query.tobytes() + pk.a_query.tobytes()\n"
            f"ref, _ = cc.read_zkey(open({src!r}, 'rb').read(), lib=lib)\n"
            "assert before == after == ref.h_query.tobytes() + ref.a_query.tobytes()\n"
            "print('ok')\n")
    env = dict(os.environ, G16_ZKEY_COPY="1", G16_NO_TORCH_PRELOAD="1", PYTHONPATH=os.pathsep.join(sys.path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_python_sources_have_no_undefined_names():
    """bench.py's per-process N > 1 path called two helpers (`exchange`, `gather`) whose definitions had
    been deleted in round 3: a NameError that only a run on that path -- never exercised by the suites --
    could show.  A coarse static check closes that class: every name that is LOADED anywhere in a module
    must be BOUND somewhere in it (any scope), or be a builtin."""
    import ast
    import builtins
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for pat in ("circom_compat_amd/*.py", "scripts/*.py", "oracle/*.py", "tests/helpers.py", "tests/golden/*.py"):
        files += sorted(glob.glob(os.path.join(root, pat)))
    assert len(files) > 10
    for path in files:
        tree = ast.parse(open(path).read(), path)
        bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for n in ast.walk(tree):
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                bound.add(n.name)
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                a = n.args
                for x in a.args + a.kwonlyargs + a.posonlyargs:
                    bound.add(x.arg)
                for x in (a.vararg, a.kwarg):
                    if x:
                        bound.add(x.arg)
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                bound.add(n.id)
            if isinstance(n, (ast.Import, ast.ImportFrom)):
                for al in n.names:
                    bound.add((al.asname or al.name).split(".")[0])
            if isinstance(n, ast.ExceptHandler) and n.name:
                bound.add(n.name)
            if isinstance(n, (ast.Global, ast.Nonlocal)):
                bound.update(n.names)
        used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
        assert not (used - bound), "%s: names used but never bound: %r" % (os.path.relpath(path, root), sorted(used - bound))


def test_shell_scripts_parse():
    """every scripts/*.sh at least parses (bash -n): the measurement scripts run once or twice per round on
    a GPU box with a minute-scale turnaround, a syntax error there costs a whole call"""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "scripts", "*.sh")))
    assert files
    for f in files:
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, "%s: %s" % (os.path.basename(f), r.stderr)
