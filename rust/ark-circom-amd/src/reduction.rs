//! `GpuCircomReduction`: `impl R1CSToQAP` (the trait `CircomReduction` implements at
//! reference src/circom/qap.rs:14-106) whose `witness_map_from_matrices` runs on the GPU.
//!
//! The trait is stateless (static methods, no `self`), so the resident state lives in a
//! thread-local cache keyed by the CONTENT of the matrices:
//!   * default (safe): every call hashes every row -- a 128-bit hash over the row lengths, the wire
//!     indices and the coefficients' raw Montgomery limbs, eight bytes per step (no `into_bigint()`:
//!     one Montgomery reduction per coefficient would cost more than the GPU witness map it guards).
//!     Equal hash and shape: the device ctx is reused; anything else: A and B are packed and uploaded
//!     again.  A matrix mutated in place, or a near-identical one reallocated at the same addresses,
//!     can therefore never meet a stale ctx (round 4's address + 64-row-sample shortcut could).
//!   * opt-in (`GpuCircomReduction::trust_unchanged_matrices(true)`, per thread): for callers that
//!     GUARANTEE the matrices behind an address are not mutated while cached (the usual prove loop
//!     over one key).  The hash is then skipped when the addresses and lengths of the outer `Vec`s and
//!     of their first rows, the shape (rows, total nnz of A and of B) and a sample of 64 rows all match
//!     -- O(rows) for the nnz count, no hashing.  `GpuCircomReduction::invalidate()` drops the cached
//!     ctx explicitly.
//! Errors: device / library failures surface as `GpuError::Library(code, message)` from
//! `GpuCircomReduction::try_witness_map`; the `R1CSToQAP` impl has only `SynthesisError` to return, so
//! it prints the library's message (`g16_last_error`) to stderr and returns
//! `SynthesisError::UnexpectedIdentity` -- never `Unsatisfiable`: an out-of-memory device is not an
//! unsatisfiable circuit.
//! The MSMs inside `ark_groth16` are not overridable through this trait -- use `GpuProver` /
//! `Groth16Gpu` for the whole proof; this impl exists for callers that only want `h`.
use std::any::TypeId;
use std::cell::RefCell;

use ark_bn254::Fr;
use ark_circom::CircomReduction;
use ark_ff::PrimeField;
use ark_groth16::r1cs_to_qap::R1CSToQAP;
use ark_poly::EvaluationDomain;
use ark_relations::r1cs::{ConstraintMatrices, ConstraintSystemRef, SynthesisError};

use crate::ffi;
use crate::pack::{self, Csr};
use crate::prover::GpuError;

pub struct GpuCircomReduction;

thread_local! {
    static TRUST_UNCHANGED: std::cell::Cell<bool> = std::cell::Cell::new(false);
}

impl GpuCircomReduction {
    /// Per-thread opt-in to the address + sample fast path (module docs): only for callers that do
    /// not mutate or replace-in-place the matrices while they are cached.
    pub fn trust_unchanged_matrices(on: bool) {
        TRUST_UNCHANGED.with(|t| t.set(on));
    }
    /// Drop this thread's cached device ctx (the next call uploads A and B again).
    pub fn invalidate() {
        CACHE.with(|c| *c.borrow_mut() = None);
    }
    /// `witness_map_from_matrices` for BN254 with the library's own error type.
    pub fn try_witness_map(
        matrices: &ConstraintMatrices<Fr>,
        num_inputs: usize,
        num_constraints: usize,
        full_assignment: &[Fr],
    ) -> Result<Vec<Fr>, GpuError> {
        gpu_witness_map(matrices, num_inputs, num_constraints, full_assignment)
    }
}

fn total_nnz(m: &ConstraintMatrices<Fr>) -> (usize, usize) {
    (m.a.iter().map(|r| r.len()).sum(), m.b.iter().map(|r| r.len()).sum())
}

fn lib_error(ctx: *const ffi::g16_ctx, st: std::os::raw::c_int) -> GpuError {
    if st == ffi::G16_ERR_DOMAIN_TOO_LARGE {
        return GpuError::Synthesis(SynthesisError::PolynomialDegreeTooLarge);
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(ffi::g16_last_error(ctx)).to_string_lossy().into_owned() };
    GpuError::Library(st, msg)
}

/// where the matrices live: outer Vec addresses / lengths and the first rows' addresses
#[derive(Clone, Copy, PartialEq, Eq)]
struct Ident {
    a_ptr: usize,
    a_len: usize,
    b_ptr: usize,
    b_len: usize,
    a_row0: usize,
    b_row0: usize,
}
fn ident_of(m: &ConstraintMatrices<Fr>) -> Ident {
    Ident {
        a_ptr: m.a.as_ptr() as usize,
        a_len: m.a.len(),
        b_ptr: m.b.as_ptr() as usize,
        b_len: m.b.len(),
        a_row0: m.a.first().map_or(0, |r| r.as_ptr() as usize),
        b_row0: m.b.first().map_or(0, |r| r.as_ptr() as usize),
    }
}

const SAMPLE_ROWS: usize = 32; // per matrix

/// copies of SAMPLE_ROWS evenly spaced rows of A and of B: (is_b, row index, row)
type RowSample = Vec<(bool, usize, Vec<(Fr, usize)>)>;
fn sample_of(m: &ConstraintMatrices<Fr>) -> RowSample {
    let mut out = Vec::new();
    for (is_b, mat) in [(false, &m.a), (true, &m.b)] {
        let n = mat.len();
        let take = SAMPLE_ROWS.min(n);
        for j in 0..take {
            let i = j * n / take;
            out.push((is_b, i, mat[i].clone()));
        }
    }
    out
}
fn sample_matches(m: &ConstraintMatrices<Fr>, s: &RowSample) -> bool {
    s.iter().all(|(is_b, i, row)| {
        let mat = if *is_b { &m.b } else { &m.a };
        mat.get(*i).map_or(false, |r| r == row)
    })
}

struct WmCtx {
    key: ((u64, u64), usize, usize, usize), // (content hash of A and B, num_constraints, num_inputs, n_vars)
    nnz: (usize, usize),
    ident: Ident,
    sample: RowSample,
    ctx: *mut ffi::g16_ctx,
    domain_size: usize,
}
impl Drop for WmCtx {
    fn drop(&mut self) {
        unsafe { ffi::g16_ctx_destroy(self.ctx) }
    }
}
thread_local! {
    static CACHE: RefCell<Option<WmCtx>> = RefCell::new(None);
}

/// 128-bit content hash (two independent multiply-rotate lanes over 64-bit words) of the rows of A
/// and B: row lengths, wire indices and the coefficients' raw Montgomery limbs
fn content_hash(m: &ConstraintMatrices<Fr>) -> (u64, u64) {
    let (mut h1, mut h2): (u64, u64) = (0x9e37_79b9_7f4a_7c15, 0xc2b2_ae3d_27d4_eb4f);
    let mut eat = |x: u64| {
        h1 = (h1 ^ x).wrapping_mul(0xff51_afd7_ed55_8ccd).rotate_left(27);
        h2 = (h2.rotate_left(31) ^ x.wrapping_mul(0x9fb2_1c65_1e98_df25)).wrapping_mul(0xc4ce_b9fe_1a85_ec53);
    };
    for mat in [&m.a, &m.b] {
        eat(mat.len() as u64);
        for row in mat.iter() {
            eat(row.len() as u64);
            for (coeff, idx) in row.iter() {
                eat(*idx as u64);
                for limb in (coeff.0).0 {
                    eat(limb); // Montgomery limbs as stored: equal elements have equal limbs
                }
            }
        }
    }
    // final avalanche of both lanes
    let fin = |mut z: u64| {
        z ^= z >> 33;
        z = z.wrapping_mul(0xff51_afd7_ed55_8ccd);
        z ^= z >> 29;
        z
    };
    (fin(h1), fin(h2 ^ h1.rotate_left(17)))
}

fn gpu_witness_map(
    matrices: &ConstraintMatrices<Fr>,
    num_inputs: usize,
    num_constraints: usize,
    full_assignment: &[Fr],
) -> Result<Vec<Fr>, GpuError> {
    let ident = ident_of(matrices);
    let shape = (num_constraints, num_inputs, full_assignment.len());
    let trust = TRUST_UNCHANGED.with(|t| t.get());
    CACHE.with(|cell| {
        let mut slot = cell.borrow_mut();
        // opt-in fast path: same place, same shape (incl. total nnz), and the row sample still reads the same
        let hit = trust
            && slot.as_ref().map_or(false, |c| {
                c.ident == ident
                    && (c.key.1, c.key.2, c.key.3) == shape
                    && c.nnz == total_nnz(matrices)
                    && sample_matches(matrices, &c.sample)
            });
        let mut key = ((0u64, 0u64), shape.0, shape.1, shape.2);
        if !hit {
            key.0 = content_hash(matrices);
            if let Some(c) = slot.as_mut() {
                if c.key == key {
                    // the same matrices at another address: keep the device ctx, re-bind the fast path
                    c.ident = ident;
                    c.sample = sample_of(matrices);
                }
            }
        }
        if !hit && slot.as_ref().map(|c| c.key) != Some(key) {
            *slot = None;
            let domain_size = (num_constraints + num_inputs).next_power_of_two();
            let (ca, cb): (Csr, Csr) = pack::matrices_to_csr(matrices);
            let (va, vb) = (ca.view(), cb.view());
            // witness-map-only ctx: a_query == NULL (include/g16_amd.h)
            let desc = ffi::g16_key_desc {
                n_vars: full_assignment.len() as u32,
                n_public: (num_inputs - 1) as u32,
                domain_size: domain_size as u32,
                a_query: std::ptr::null(),
                b_g1_query: std::ptr::null(),
                b_g2_query: std::ptr::null(),
                l_query: std::ptr::null(),
                h_query: std::ptr::null(),
                alpha_g1: [0; 64],
                beta_g1: [0; 64],
                delta_g1: [0; 64],
                beta_g2: [0; 128],
                delta_g2: [0; 128],
            };
            let mut ctx: *mut ffi::g16_ctx = std::ptr::null_mut();
            let st = unsafe { ffi::g16_ctx_create(&desc, &va, &vb, num_constraints as u32, std::ptr::null(), &mut ctx) };
            if st != ffi::G16_OK {
                return Err(lib_error(std::ptr::null(), st)); // create errors live in the library's global slot
            }
            *slot = Some(WmCtx { key, nnz: total_nnz(matrices), ident, sample: sample_of(matrices), ctx, domain_size });
        }
        let c = slot.as_ref().unwrap();
        let w = pack::fr_vec_words(full_assignment);
        let mut h = vec![0u64; 4 * c.domain_size];
        let st = unsafe { ffi::g16_witness_map(c.ctx, w.as_ptr(), full_assignment.len(), h.as_mut_ptr()) };
        if st != ffi::G16_OK {
            return Err(lib_error(c.ctx, st));
        }
        Ok(h.chunks_exact(4).map(|l| Fr::new_unchecked(ark_ff::BigInt([l[0], l[1], l[2], l[3]]))).collect())
    })
}

impl R1CSToQAP for GpuCircomReduction {
    #[allow(clippy::type_complexity)]
    fn instance_map_with_evaluation<F: PrimeField, D: EvaluationDomain<F>>(
        cs: ConstraintSystemRef<F>,
        t: &F,
    ) -> Result<(Vec<F>, Vec<F>, Vec<F>, F, usize, usize), SynthesisError> {
        // key generation side: unchanged (CircomReduction delegates to LibsnarkReduction, qap.rs:16-21)
        CircomReduction::instance_map_with_evaluation::<F, D>(cs, t)
    }

    fn witness_map_from_matrices<F: PrimeField, D: EvaluationDomain<F>>(
        matrices: &ConstraintMatrices<F>,
        num_inputs: usize,
        num_constraints: usize,
        full_assignment: &[F],
    ) -> Result<Vec<F>, SynthesisError> {
        if TypeId::of::<F>() == TypeId::of::<Fr>() {
            // F is ark_bn254::Fr: the casts below are identity casts
            let m = unsafe { &*(matrices as *const ConstraintMatrices<F> as *const ConstraintMatrices<Fr>) };
            let w = unsafe { std::slice::from_raw_parts(full_assignment.as_ptr() as *const Fr, full_assignment.len()) };
            let h = gpu_witness_map(m, num_inputs, num_constraints, w).map_err(|e| match e {
                GpuError::Synthesis(s) => s,
                GpuError::Library(code, msg) => {
                    // the trait offers no "device failure" variant; say what happened where a log will keep it
                    eprintln!("ark-circom-amd: libg16_amd failed in witness_map_from_matrices (status {code}): {msg}");
                    SynthesisError::UnexpectedIdentity
                }
            })?;
            let h = std::mem::ManuallyDrop::new(h);
            // Vec<Fr> -> Vec<F>, same type
            return Ok(unsafe { Vec::from_raw_parts(h.as_ptr() as *mut F, h.len(), h.capacity()) });
        }
        // any other field is not this library's business: the reference implementation
        CircomReduction::witness_map_from_matrices::<F, D>(matrices, num_inputs, num_constraints, full_assignment)
    }

    fn h_query_scalars<F: PrimeField, D: EvaluationDomain<F>>(
        max_power: usize,
        t: F,
        zt: F,
        delta_inverse: F,
    ) -> Result<Vec<F>, SynthesisError> {
        CircomReduction::h_query_scalars::<F, D>(max_power, t, zt, delta_inverse)
    }
}
