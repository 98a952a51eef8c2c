"""rust/ark-circom-amd cannot be compiled here (no Rust toolchain): check mechanically what can be --
its `extern "C"` block against the C headers (same function set, same arity, same pointer-ness per
argument) and its #[repr(C)] structs against the C structs (same field names in the same order)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "ark-circom-amd")


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"#ifdef G16_DEBUG_ABI.*?#endif", " ", text, flags=re.S)   # measurement builds only
    return re.sub(r"//[^\n]*", " ", text)


def c_functions():
    out = {}
    for h in ("g16_amd.h", "g16_loaders.h"):
        text = _strip_c(open(os.path.join(ROOT, "include", h)).read())
        text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(g16_\w+)\s*\(([^;{]*?)\)\s*;", text):
            name, args = m.group(2), m.group(3).strip()
            params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
            out[name] = ["*" in p or "[" in p for p in params]
    return out


def rust_functions():
    text = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    text = re.sub(r"//[^\n]*", " ", text)
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (g16_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        params = [a.strip() for a in args.split(",") if a.strip()]
        out[m.group(1)] = [p.split(":", 1)[1].strip().startswith("*") for p in params]
    return out


def test_extern_c_block_matches_the_headers():
    c, r = c_functions(), rust_functions()
    assert set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert len(c[name]) == len(r[name]), (name, c[name], r[name])
        assert c[name] == r[name], f"{name}: pointer / value arguments differ: C {c[name]} vs Rust {r[name]}"
    # and the ctypes binding covers the same set (tests/test_loaders_abi.py checks it against the .so)
    from circom_compat_amd import _binding
    assert set(_binding.ABI_SYMBOLS) == set(c)


def _c_struct_fields(name):
    for h in ("g16_amd.h", "g16_loaders.h"):
        text = _strip_c(open(os.path.join(ROOT, "include", h)).read())
        structs = {m.group(2): m.group(1) for m in re.finditer(r"typedef\s+struct\s*\{([^{}]*)\}\s*(\w+)\s*;", text)}
        if name in structs:
            fields = []
            for decl in structs[name].split(";"):
                decl = decl.strip()
                if not decl:
                    continue
                for part in decl.split(","):
                    ident = re.findall(r"([A-Za-z_]\w*)\s*(?:\[[^\]]*\])?\s*$", part.strip())
                    fields.append(ident[0])
            return fields
    raise KeyError(name)


def _rust_struct_fields(name):
    text = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    m = re.search(r"#\[repr\(C\)\][^{]*?pub struct " + name + r"\s*\{(.*?)\n\}", text, flags=re.S)
    return re.findall(r"pub (\w+)\s*:", m.group(1))


def test_repr_c_structs_match_the_headers():
    for name in ("g16_csr", "g16_key_desc", "g16_options", "g16_vk_desc", "g16_zkey_header", "g16_matrices", "g16_r1cs_header"):
        assert _rust_struct_fields(name) == _c_struct_fields(name), name


def test_crate_is_complete_source():
    for rel in ("Cargo.toml", "build.rs", "README.md", "src/lib.rs", "src/ffi.rs", "src/pack.rs", "src/prover.rs",
                "src/reduction.rs", "tests/zkey.rs", "benches/groth16.rs"):
        assert os.path.getsize(os.path.join(CRATE, rel)) > 200, rel
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    for name in ("CircomConfig", "CircomBuilder", "CircomCircuit", "CircomReduction", "read_zkey", "GpuProver",
                 "GpuCircomReduction", "Groth16Gpu"):
        assert name in lib, name
    assert "impl R1CSToQAP for GpuCircomReduction" in open(os.path.join(CRATE, "src", "reduction.rs")).read()
    t = open(os.path.join(CRATE, "tests", "zkey.rs")).read()
    assert "verify_proof_with_zkey_with_r1cs" in t and "verify_proof_with_zkey_without_r1cs" in t
