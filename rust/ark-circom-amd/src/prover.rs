//! `GpuProver`: the device-resident `(ProvingKey<Bn254>, ConstraintMatrices<Fr>)` pair that
//! `create_proof_with_reduction_and_matrices` borrows on every call in the reference
//! (benches/groth16.rs:52-60, src/zkey.rs:903-911), uploaded and precomputed once.
use std::ffi::CStr;
use std::os::raw::c_int;

use ark_bn254::{Bn254, Fr};
use ark_groth16::{Proof, ProvingKey};
use ark_relations::r1cs::{ConstraintMatrices, SynthesisError};
use ark_std::rand::Rng;
use ark_std::UniformRand;

use crate::ffi;
use crate::pack::{self, Csr};

/// Which `R1CSToQAP` the key was generated for (README.md:69-74 of the reference: never mix them).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Reduction {
    /// `ark_circom::CircomReduction`: snarkjs keys (`.zkey`)
    Circom,
    /// `ark_groth16::LibsnarkReduction`: keys from `Groth16::<Bn254>::generate_random_parameters_with_reduction`
    Libsnark,
}

#[derive(Debug)]
pub enum GpuError {
    /// `SynthesisError::PolynomialDegreeTooLarge` and friends, as the CPU path reports them
    Synthesis(SynthesisError),
    /// anything else the library reports (status code, message)
    Library(i32, String),
}
impl std::fmt::Display for GpuError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            GpuError::Synthesis(s) => write!(f, "{s}"),
            GpuError::Library(code, msg) => write!(f, "libg16_amd status {code}: {msg}"),
        }
    }
}
impl std::error::Error for GpuError {}
impl From<SynthesisError> for GpuError {
    fn from(e: SynthesisError) -> Self {
        GpuError::Synthesis(e)
    }
}
// Deliberately NO blanket `impl From<GpuError> for SynthesisError`: no SynthesisError variant means
// "the device failed", and rounds 2-4 squeezed a HIP out-of-memory into `Unsatisfiable`.  The typed
// entry points (`GpuProver::*`, `Groth16Gpu::try_*`) return `GpuError`; the places that keep the
// reference's `SynthesisError` signature (`Groth16Gpu::{prove, create_proof_with_reduction_and_matrices}`
// in lib.rs, the `R1CSToQAP` impl in reduction.rs) convert explicitly: they log the library's status and
// message first and return `UnexpectedIdentity`.

pub struct GpuProver {
    ctx: *mut ffi::g16_ctx,
    n_vars: usize,
    num_inputs: usize,
    num_constraints: usize,
}
// one proof in flight per ctx (the C ABI's contract); moving the handle between threads is fine
unsafe impl Send for GpuProver {}

fn last_error(ctx: *const ffi::g16_ctx) -> String {
    unsafe { CStr::from_ptr(ffi::g16_last_error(ctx)).to_string_lossy().into_owned() }
}
fn check(ctx: *const ffi::g16_ctx, st: c_int) -> Result<(), GpuError> {
    match st {
        ffi::G16_OK => Ok(()),
        ffi::G16_ERR_DOMAIN_TOO_LARGE => Err(GpuError::Synthesis(SynthesisError::PolynomialDegreeTooLarge)),
        _ => Err(GpuError::Library(st, last_error(ctx))),
    }
}

/// How `with_devices_sharded` cuts the MSMs over the devices (`G16_SHARD_*`, include/g16_amd.h).
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum Shard {
    Auto,
    Points,
    Buckets,
}

impl GpuProver {
    /// The inputs `read_zkey` returns (src/zkey.rs:53-60), on one GPU.
    pub fn new(pk: &ProvingKey<Bn254>, matrices: &ConstraintMatrices<Fr>) -> Result<Self, GpuError> {
        Self::with_devices(pk, matrices, &[0], Reduction::Circom)
    }

    /// `devices.len() > 1`: ONE prover sharded over several GPUs inside the library
    /// (`g16_ctx_create_multi`): point-range MSM shards, distributed witness map, peer copies
    /// over xGMI.  The call surface does not change.
    pub fn with_devices(
        pk: &ProvingKey<Bn254>,
        matrices: &ConstraintMatrices<Fr>,
        devices: &[i32],
        reduction: Reduction,
    ) -> Result<Self, GpuError> {
        Self::with_devices_sharded(pk, matrices, devices, reduction, Shard::Auto)
    }

    /// The same with the cut of the MSMs named (`g16_options.shard`): `Shard::Points` = every device
    /// holds 1/n of the key, `Shard::Buckets` = every device holds all points of the witness queries
    /// and the single-GPU window and works on 1/n of the sorted bucket list (same time per rank,
    /// DESIGN.md section 7); `Shard::Auto` = points.
    pub fn with_devices_sharded(
        pk: &ProvingKey<Bn254>,
        matrices: &ConstraintMatrices<Fr>,
        devices: &[i32],
        reduction: Reduction,
        shard: Shard,
    ) -> Result<Self, GpuError> {
        let ids: Vec<c_int> = devices.iter().map(|d| *d as c_int).collect();
        Self::create_with(pk, matrices, reduction, shard, |opt| opt, |key, va, vb, m, opt, ctx| unsafe {
            ffi::g16_ctx_create_multi(key, va, vb, m, ids.as_ptr(), ids.len() as c_int, opt, ctx)
        })
    }

    /// One RANK of a prover sharded over `world` processes, one GPU each, with the collectives issued by
    /// the library through RCCL (`g16_dist_attach_rccl`, include/g16_amd.h): `nccl_comm` is the
    /// `ncclComm_t` this process created over the same `world` ranks with its own RCCL binding
    /// (ncclGetUniqueId on rank 0, the id shared by the host's own means, ncclCommInitRank everywhere).
    /// `prove_dist` is then one whole proof per rank: every rank gets the same `Proof`.
    ///
    /// # Safety
    /// `nccl_comm` must be a live communicator of exactly `world` ranks in which this process is `rank`,
    /// created on `device`; it must outlive the prover (the library never destroys it).
    pub unsafe fn with_rccl(
        pk: &ProvingKey<Bn254>,
        matrices: &ConstraintMatrices<Fr>,
        device: i32,
        rank: i32,
        world: i32,
        nccl_comm: *mut std::os::raw::c_void,
        shard: Shard,
    ) -> Result<Self, GpuError> {
        let me = Self::create_with(
            pk,
            matrices,
            Reduction::Circom,
            shard,
            |mut opt| {
                opt.device = device as c_int;
                opt.rank = rank as c_int;
                opt.world = world as c_int;
                opt.dist_wm = 1;
                opt
            },
            |key, va, vb, m, opt, ctx| unsafe { ffi::g16_ctx_create(key, va, vb, m, opt, ctx) },
        )?;
        check(me.ctx, ffi::g16_dist_attach_rccl(me.ctx, nccl_comm))?;
        Ok(me)
    }

    /// ranks of the attached communicator (0: none)
    pub fn rccl_ranks(&self) -> usize {
        unsafe { ffi::g16_dist_rccl_ranks(self.ctx) as usize }
    }

    /// One sharded proof of this rank (`g16_prove_dist`): phase 1 -> ncclAllToAll -> phase 2 -> ncclAllToAll ->
    /// phase 3 -> ncclAllGather of the 1 KiB records -> finish.  `w_dev`: the full assignment in device
    /// memory (`4 * n_vars` u64, Montgomery), e.g. `g16_witness_buffer` after `g16_witness_upload`.
    ///
    /// # Safety
    /// `w_dev` must point at `n_vars` field elements in the memory of this prover's device.
    pub unsafe fn prove_dist(&mut self, r: Fr, s: Fr, w_dev: *const std::os::raw::c_void) -> Result<Proof<Bn254>, GpuError> {
        let mut raw = [0u8; ffi::G16_PROOF_BYTES];
        let (rw, sw) = (pack::fr_words(&r), pack::fr_words(&s));
        check(self.ctx, ffi::g16_prove_dist(self.ctx, rw.as_ptr(), sw.as_ptr(), w_dev, self.n_vars, raw.as_mut_ptr()))?;
        Ok(pack::unpack_proof(&raw))
    }

    /// packs the key and the matrices once and hands them to `make` (g16_ctx_create / _multi)
    fn create_with(
        pk: &ProvingKey<Bn254>,
        matrices: &ConstraintMatrices<Fr>,
        reduction: Reduction,
        shard: Shard,
        tune: impl FnOnce(ffi::g16_options) -> ffi::g16_options,
        make: impl FnOnce(&ffi::g16_key_desc, &ffi::g16_csr, &ffi::g16_csr, u32, &ffi::g16_options, &mut *mut ffi::g16_ctx) -> c_int,
    ) -> Result<Self, GpuError> {
        let n_vars = pk.a_query.len();
        let n_public = pk.vk.gamma_abc_g1.len() - 1;
        let a = pack::pack_g1_vec(&pk.a_query);
        let b1 = pack::pack_g1_vec(&pk.b_g1_query);
        let b2 = pack::pack_g2_vec(&pk.b_g2_query);
        let l = pack::pack_g1_vec(&pk.l_query);
        // LibsnarkReduction's H query has domain_size - 1 points: pad with the point at infinity
        let mut h = pack::pack_g1_vec(&pk.h_query);
        let need = matrices.num_constraints + matrices.num_instance_variables;
        let domain_size = need.next_power_of_two();
        h.resize(64 * domain_size, 0);
        let mut key = ffi::g16_key_desc {
            n_vars: n_vars as u32,
            n_public: n_public as u32,
            domain_size: domain_size as u32,
            a_query: a.as_ptr(),
            b_g1_query: b1.as_ptr(),
            b_g2_query: b2.as_ptr(),
            l_query: l.as_ptr(),
            h_query: h.as_ptr(),
            alpha_g1: [0; 64],
            beta_g1: [0; 64],
            delta_g1: [0; 64],
            beta_g2: [0; 128],
            delta_g2: [0; 128],
        };
        pack::pack_g1(&pk.vk.alpha_g1, &mut key.alpha_g1);
        pack::pack_g1(&pk.beta_g1, &mut key.beta_g1);
        pack::pack_g1(&pk.delta_g1, &mut key.delta_g1);
        pack::pack_g2(&pk.vk.beta_g2, &mut key.beta_g2);
        pack::pack_g2(&pk.vk.delta_g2, &mut key.delta_g2);
        let (ca, cb): (Csr, Csr) = pack::matrices_to_csr(matrices);
        let (va, vb) = (ca.view(), cb.view());
        let opt = tune(ffi::g16_options {
            reduction: if reduction == Reduction::Libsnark { ffi::G16_REDUCTION_LIBSNARK } else { ffi::G16_REDUCTION_CIRCOM },
            shard: match shard {
                Shard::Auto => ffi::G16_SHARD_AUTO,
                Shard::Points => ffi::G16_SHARD_POINTS,
                Shard::Buckets => ffi::G16_SHARD_BUCKETS,
            },
            ..Default::default()
        });
        let mut ctx: *mut ffi::g16_ctx = std::ptr::null_mut();
        let st = make(&key, &va, &vb, matrices.num_constraints as u32, &opt, &mut ctx);
        check(std::ptr::null(), st)?;
        Ok(GpuProver { ctx, n_vars, num_inputs: matrices.num_instance_variables, num_constraints: matrices.num_constraints })
    }

    pub fn num_inputs(&self) -> usize {
        self.num_inputs
    }
    pub fn num_constraints(&self) -> usize {
        self.num_constraints
    }

    /// `Groth16::<Bn254, CircomReduction>::create_proof_with_reduction_and_matrices` with
    /// `(pk, matrices, num_inputs, num_constraints)` taken from `self`.
    pub fn create_proof(&mut self, r: Fr, s: Fr, full_assignment: &[Fr]) -> Result<Proof<Bn254>, GpuError> {
        if full_assignment.len() != self.n_vars {
            return Err(GpuError::Synthesis(SynthesisError::MalformedVerifyingKey));
        }
        let mut raw = [0u8; ffi::G16_PROOF_BYTES];
        let (rw, sw) = (pack::fr_words(&r), pack::fr_words(&s));
        // the witness goes through the ctx's page-locked staging buffer: H2D at PCIe line rate
        let st = unsafe {
            let host = ffi::g16_witness_host_buffer(self.ctx) as *mut u64;
            if host.is_null() {
                let w = pack::fr_vec_words(full_assignment);
                ffi::g16_prove(self.ctx, rw.as_ptr(), sw.as_ptr(), w.as_ptr(), self.n_vars, raw.as_mut_ptr())
            } else {
                pack::fr_write_words(full_assignment, std::slice::from_raw_parts_mut(host, 4 * self.n_vars));
                ffi::g16_prove(self.ctx, rw.as_ptr(), sw.as_ptr(), host, self.n_vars, raw.as_mut_ptr())
            }
        };
        check(self.ctx, st)?;
        Ok(pack::unpack_proof(&raw))
    }

    /// `SNARK::prove(&pk, circuit, rng)` shape (src/zkey.rs:866): r, s from the rng.
    pub fn prove_with_rng<R: Rng>(&mut self, full_assignment: &[Fr], rng: &mut R) -> Result<Proof<Bn254>, GpuError> {
        let r = Fr::rand(rng);
        let s = Fr::rand(rng);
        self.create_proof(r, s, full_assignment)
    }

    /// `CircomReduction::witness_map_from_matrices` (src/circom/qap.rs:23-88) on the resident matrices.
    pub fn witness_map(&mut self, full_assignment: &[Fr], domain_size: usize) -> Result<Vec<Fr>, GpuError> {
        let w = pack::fr_vec_words(full_assignment);
        let mut h = vec![0u64; 4 * domain_size];
        check(self.ctx, unsafe { ffi::g16_witness_map(self.ctx, w.as_ptr(), full_assignment.len(), h.as_mut_ptr()) })?;
        Ok(h.chunks_exact(4)
            .map(|c| Fr::new_unchecked(ark_ff::BigInt([c[0], c[1], c[2], c[3]])))
            .collect())
    }
}

impl Drop for GpuProver {
    fn drop(&mut self) {
        unsafe { ffi::g16_ctx_destroy(self.ctx) }
    }
}
