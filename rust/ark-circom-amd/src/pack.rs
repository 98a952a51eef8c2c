//! Repacking between arkworks types and the packed forms of the C ABI.
//!
//! `ark_ff::Fp256<MontBackend<_, 4>>` keeps its value as a `BigInt<4>` in Montgomery form: `x.0.0`
//! is exactly the 4 x u64 the ABI wants -- the convention `deserialize_field` relies on in the
//! reference (src/zkey.rs:327-332).  `ark_ec::short_weierstrass::Affine { x, y, infinity }` is
//! `repr(Rust)` (72 / 136 bytes), so points are repacked to the zkey encoding: x | y, all-zero =
//! infinity (src/zkey.rs:340-360).
use ark_bn254::{Fq, Fq2, Fr, G1Affine, G2Affine};
use ark_ff::BigInt;
use ark_groth16::Proof;
use ark_relations::r1cs::ConstraintMatrices;

#[inline]
fn fq_words(x: &Fq) -> [u64; 4] {
    x.0 .0
}
#[inline]
pub fn fr_words(x: &Fr) -> [u64; 4] {
    x.0 .0
}
#[inline]
fn put(words: &[u64; 4], out: &mut [u8]) {
    for (i, w) in words.iter().enumerate() {
        out[8 * i..8 * i + 8].copy_from_slice(&w.to_le_bytes());
    }
}
#[inline]
fn get(b: &[u8]) -> Fq {
    let mut l = [0u64; 4];
    for (i, w) in l.iter_mut().enumerate() {
        *w = u64::from_le_bytes(b[8 * i..8 * i + 8].try_into().unwrap());
    }
    Fq::new_unchecked(BigInt(l)) // already Montgomery
}

pub fn pack_g1(p: &G1Affine, out: &mut [u8]) {
    debug_assert_eq!(out.len(), 64);
    if p.infinity {
        out.fill(0);
        return;
    }
    put(&fq_words(&p.x), &mut out[0..32]);
    put(&fq_words(&p.y), &mut out[32..64]);
}

/// x.c0 | x.c1 | y.c0 | y.c1
pub fn pack_g2(p: &G2Affine, out: &mut [u8]) {
    debug_assert_eq!(out.len(), 128);
    if p.infinity {
        out.fill(0);
        return;
    }
    put(&fq_words(&p.x.c0), &mut out[0..32]);
    put(&fq_words(&p.x.c1), &mut out[32..64]);
    put(&fq_words(&p.y.c0), &mut out[64..96]);
    put(&fq_words(&p.y.c1), &mut out[96..128]);
}

pub fn pack_g1_vec(v: &[G1Affine]) -> Vec<u8> {
    let mut b = vec![0u8; 64 * v.len()];
    for (p, o) in v.iter().zip(b.chunks_exact_mut(64)) {
        pack_g1(p, o);
    }
    b
}
pub fn pack_g2_vec(v: &[G2Affine]) -> Vec<u8> {
    let mut b = vec![0u8; 128 * v.len()];
    for (p, o) in v.iter().zip(b.chunks_exact_mut(128)) {
        pack_g2(p, o);
    }
    b
}

pub fn unpack_g1(b: &[u8]) -> G1Affine {
    if b.iter().all(|&x| x == 0) {
        return G1Affine::identity();
    }
    G1Affine::new_unchecked(get(&b[0..32]), get(&b[32..64]))
}
pub fn unpack_g2(b: &[u8]) -> G2Affine {
    if b.iter().all(|&x| x == 0) {
        return G2Affine::identity();
    }
    G2Affine::new_unchecked(
        Fq2::new(get(&b[0..32]), get(&b[32..64])),
        Fq2::new(get(&b[64..96]), get(&b[96..128])),
    )
}

/// A(64) | B(128) | C(64) -> `ark_groth16::Proof<Bn254>`
pub fn unpack_proof(raw: &[u8; 256]) -> Proof<ark_bn254::Bn254> {
    Proof {
        a: unpack_g1(&raw[0..64]),
        b: unpack_g2(&raw[64..192]),
        c: unpack_g1(&raw[192..256]),
    }
}

/// `&[Fr]` -> the contiguous 4 x u64 Montgomery words the ABI reads.  `Fr` is a transparent wrapper
/// chain around `[u64; 4]` in arkworks 0.5, but that is not a documented layout guarantee, so copy.
pub fn fr_vec_words(v: &[Fr]) -> Vec<u64> {
    let mut out = Vec::with_capacity(4 * v.len());
    for x in v {
        out.extend_from_slice(&fr_words(x));
    }
    out
}
/// the same, straight into a caller-provided (e.g. page-locked) buffer
pub fn fr_write_words(v: &[Fr], out: &mut [u64]) {
    for (x, o) in v.iter().zip(out.chunks_exact_mut(4)) {
        o.copy_from_slice(&fr_words(x));
    }
}

/// `ConstraintMatrices::{a, b}` rows (`Vec<Vec<(Fr, usize)>>`, src/zkey.rs:165-194) -> CSR
pub struct Csr {
    pub row_ptr: Vec<u32>,
    pub col: Vec<u32>,
    pub coeff: Vec<u64>,
}
impl Csr {
    pub fn from_rows(rows: &[Vec<(Fr, usize)>]) -> Self {
        let nnz: usize = rows.iter().map(|r| r.len()).sum();
        let mut c = Csr { row_ptr: Vec::with_capacity(rows.len() + 1), col: Vec::with_capacity(nnz), coeff: Vec::with_capacity(4 * nnz) };
        c.row_ptr.push(0);
        for row in rows {
            for (cf, idx) in row {
                c.col.push(*idx as u32);
                c.coeff.extend_from_slice(&fr_words(cf));
            }
            c.row_ptr.push(c.col.len() as u32);
        }
        c
    }
    pub fn view(&self) -> crate::ffi::g16_csr {
        crate::ffi::g16_csr { row_ptr: self.row_ptr.as_ptr(), col: self.col.as_ptr(), coeff: self.coeff.as_ptr(), nnz: self.col.len() as u64 }
    }
}

pub fn matrices_to_csr(m: &ConstraintMatrices<Fr>) -> (Csr, Csr) {
    (Csr::from_rows(&m.a), Csr::from_rows(&m.b))
}
