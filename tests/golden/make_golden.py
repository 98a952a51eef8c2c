"""Generates tests/golden/reference_vectors.json from the READ-ONLY reference checkout.

Run in the build container only (`python tests/golden/make_golden.py /root/reference`); the GPU box
never reads /root/reference.  It extracts DATA, not code: the literal byte arrays the reference's
own unit tests compare against (src/zkey.rs:398-432 snarkjs dumps of Fq one / G1 / G2 generators,
src/zkey.rs:551-762 every point of test.zkey, src/circom/r1cs_reader.rs:259-309 the hand-written
.r1cs sample) plus the expectations those tests assert."""
import json
import os
import re
import sys


def byte_arrays(text):
    """all `[ n, n, ... ]` literals of u8 values, in source order"""
    out = []
    for m in re.finditer(r"\[\s*((?:\d{1,3}\s*,\s*)+\d{1,3}\s*,?\s*)\]", text):
        vals = [int(x) for x in re.findall(r"\d+", m.group(1))]
        if all(v < 256 for v in vals) and len(vals) in (32, 64, 128):
            out.append(vals)
    return out


def main(ref):
    z = open(os.path.join(ref, "src/zkey.rs")).read()
    tests = z[z.index("#[cfg(test)]"):]
    fq_buf = byte_arrays(tests[tests.index("fn fq_buf"):tests.index("fn g1_buf")])[0]
    g1_buf = byte_arrays(tests[tests.index("fn g1_buf"):tests.index("fn g2_buf")])[0]
    g2_buf = byte_arrays(tests[tests.index("fn g2_buf"):tests.index("fn g1_one")])[0]
    dk = tests[tests.index("fn deser_key"):tests.index("fn deser_vk")]
    sections = {}
    marks = [("ic", "// Check IC"), ("a_query", "// Check A Query"), ("b_g1_query", "params.a_query"),
             ("b_g2_query", "params.b_g1_query"), ("l_query", "// Check L Query"),
             ("h_query", "// Check H Query"), ("end", "params.h_query")]
    for (name, start), (_, stop) in zip(marks[:-1], marks[1:]):
        seg = dk[dk.index(start):dk.index(stop, dk.index(start) + len(start))]
        sections[name] = byte_arrays(seg)
    r = open(os.path.join(ref, "src/circom/r1cs_reader.rs")).read()
    hexblob = re.search(r'hex_literal::hex!\(\s*"(.*?)"\s*\)', r, re.S).group(1)
    r1cs_sample = re.sub(r"\s+", "", hexblob)
    wc = open(os.path.join(ref, "src/witness/witness_calculator.rs")).read()
    out = {
        "source": "arkworks-rs/circom-compat snapshot 2025-03-01: src/zkey.rs tests, src/circom/r1cs_reader.rs tests",
        "fq_one_mont": fq_buf, "g1_generator": g1_buf, "g2_generator": g2_buf,
        "test_zkey": sections,
        "test_zkey_header": {"n_vars": 4, "n_public": 1, "domain_size": 4, "power": 2},
        "r1cs_sample_hex": r1cs_sample,
        "r1cs_sample_expect": {"version": 1, "field_size": 32, "n_wires": 7, "n_pub_out": 1, "n_pub_in": 2,
                               "n_prv_in": 3, "n_labels": 0x03e8, "n_constraints": 3,
                               "c0_a_len": 2, "c0_a0": [5, 3], "c2_b0": [0, 6], "c1_c_len": 0,
                               "wire_mapping_len": 7, "wire_mapping_1": 3},
        "mycircuit_witness": ["1", "33", "3", "11"],
    }
    # sanity: counts match the fixture's sizes (n_vars 4, n_public 1, domain 4)
    assert [len(sections[k]) for k in ("ic", "a_query", "b_g1_query", "b_g2_query", "l_query", "h_query")] == \
        [2, 4, 4, 4, 2, 4], {k: len(v) for k, v in sections.items()}
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "reference_vectors.json")
    json.dump(out, open(dst, "w"), indent=0)
    print("wrote", dst)
    # binary FIXTURES (data, not code) the reference's tests and bench read from test-vectors/;
    # complex-circuit-10000-10000.r1cs is the default workload of benches/groth16.rs:87-108 (its
    # .zkey is not in the snapshot: the tests mint a trapdoor key for it)
    import shutil
    for rel in ("test-vectors/test.zkey", "test-vectors/mycircuit.r1cs", "test-vectors/circuit2.r1cs",
                "test-vectors/complex-circuit/complex-circuit-10000-10000.r1cs",
                "test-vectors/complex-circuit/input.json"):
        src = os.path.join(ref, rel)
        name = os.path.basename(rel) if "input.json" not in rel else "complex-circuit-input.json"
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(here, name))
            print("copied", rel)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
