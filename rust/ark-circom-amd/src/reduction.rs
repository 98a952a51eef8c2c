//! `GpuCircomReduction`: `impl R1CSToQAP` (the trait `CircomReduction` implements at
//! reference src/circom/qap.rs:14-106) whose `witness_map_from_matrices` runs on the GPU.
//!
//! The trait is stateless (static methods, no `self`), so the resident state lives in a
//! thread-local cache keyed by the matrices' CONTENT (shape + a 64-bit FNV-1a hash over every row's
//! coefficients and indices; an address would be reused by a freed-and-reallocated `Vec` of the
//! same shape and silently pair a stale device ctx with new matrices): the first call uploads A
//! and B, later calls with equal matrices only upload the witness.  Hashing is one pass over nnz
//! (coefficient, index) pairs -- cheaper than the pack + upload it saves.  The MSMs inside
//! `ark_groth16` are not overridable through this trait -- use `GpuProver` / `Groth16Gpu` for the
//! whole proof; this impl exists for callers that only want `h`.
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

pub struct GpuCircomReduction;

struct WmCtx {
    key: (u64, usize, usize, usize), // (content hash of A and B, num_constraints, num_inputs, n_vars)
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

/// FNV-1a over the rows of A and B: row lengths, wire indices and the coefficients' limbs
fn content_hash(m: &ConstraintMatrices<Fr>) -> u64 {
    let mut h: u64 = 0xcbf29ce484222325;
    let mut eat = |x: u64| {
        for b in x.to_le_bytes() {
            h ^= b as u64;
            h = h.wrapping_mul(0x100000001b3);
        }
    };
    for mat in [&m.a, &m.b] {
        eat(mat.len() as u64);
        for row in mat.iter() {
            eat(row.len() as u64);
            for (coeff, idx) in row.iter() {
                eat(*idx as u64);
                for limb in coeff.into_bigint().0 {
                    eat(limb);
                }
            }
        }
    }
    h
}

fn gpu_witness_map(
    matrices: &ConstraintMatrices<Fr>,
    num_inputs: usize,
    num_constraints: usize,
    full_assignment: &[Fr],
) -> Result<Vec<Fr>, SynthesisError> {
    let key = (content_hash(matrices), num_constraints, num_inputs, full_assignment.len());
    CACHE.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.as_ref().map(|c| c.key) != Some(key) {
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
            match st {
                ffi::G16_OK => {}
                ffi::G16_ERR_DOMAIN_TOO_LARGE => return Err(SynthesisError::PolynomialDegreeTooLarge),
                _ => return Err(SynthesisError::Unsatisfiable),
            }
            *slot = Some(WmCtx { key, ctx, domain_size });
        }
        let c = slot.as_ref().unwrap();
        let w = pack::fr_vec_words(full_assignment);
        let mut h = vec![0u64; 4 * c.domain_size];
        let st = unsafe { ffi::g16_witness_map(c.ctx, w.as_ptr(), full_assignment.len(), h.as_mut_ptr()) };
        if st != ffi::G16_OK {
            return Err(SynthesisError::Unsatisfiable);
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
            let h = gpu_witness_map(m, num_inputs, num_constraints, w)?;
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
