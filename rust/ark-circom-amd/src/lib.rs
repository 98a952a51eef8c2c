//! ark-circom-amd -- MI355X drop-in for the Groth16 proving path of `ark-circom` 0.5.
//!
//! Everything that is not the proving hot path is the reference crate, re-exported unchanged:
//! `CircomConfig`, `CircomBuilder`, `CircomCircuit`, `CircomReduction`, `WitnessCalculator`,
//! `read_zkey`.  What changes is who computes the proof:
//!
//! ```ignore
//! use ark_circom_amd::{read_zkey, CircomBuilder, CircomConfig, GpuProver, Groth16Gpu};
//! let (params, matrices) = read_zkey(&mut File::open("circuit.zkey")?)?;     // unchanged
//! let mut prover = GpuProver::new(&params, &matrices)?;                       // once per key
//! let circom = builder.build()?;                                              // unchanged (WASM witness)
//! let inputs = circom.get_public_inputs().unwrap();
//! let proof = Groth16Gpu::prove(&mut prover, circom, &mut rng)?;              // was Groth16::<Bn254, CircomReduction>::prove(&params, circom, &mut rng)
//! assert!(Groth16::<Bn254>::verify_with_processed_vk(&pvk, &inputs, &proof)?); // unchanged
//! ```
//!
//! The proof is the same group elements the CPU path produces for the same `(pk, r, s, w)`
//! (bit-identical `Proof<Bn254>`), so verification, Solidity calldata (`ark_circom::ethereum`) and
//! serialisation need no change.
pub mod ffi;
pub mod pack;
mod prover;
mod reduction;
mod verify;

pub use ark_circom::{circom, read_zkey, CircomBuilder, CircomCircuit, CircomConfig, CircomReduction, Wasm, WitnessCalculator};
pub use prover::{GpuError, GpuProver, Reduction, Shard};
pub use reduction::GpuCircomReduction;
pub use verify::verify_batch;

use ark_bn254::{Bn254, Fr};
use ark_groth16::Proof;
use ark_relations::r1cs::SynthesisError;
use ark_std::rand::Rng;
use ark_std::UniformRand;

/// The two entry points of `Groth16::<Bn254, CircomReduction>` that sit on the proving path, with the
/// reference's argument meaning AND return type (benches/groth16.rs:52-60, src/zkey.rs:866:
/// `Result<Proof<Bn254>, SynthesisError>`), so callers that use `?` into `SynthesisError` or match on it
/// keep compiling.  A device failure has no `SynthesisError` variant: like the `R1CSToQAP` impl in
/// reduction.rs, these log the library's status and message to stderr and return
/// `SynthesisError::UnexpectedIdentity` (never `Unsatisfiable`).  The `try_*` variants return the typed
/// `GpuError` (`Synthesis(..)` for everything the CPU path can report, `Library(code, message)` for
/// device failures) for callers that want to tell the two apart.
pub struct Groth16Gpu;

/// the documented lossy conversion of the reference-shaped entry points
fn to_synthesis(e: GpuError, site: &str) -> SynthesisError {
    match e {
        GpuError::Synthesis(s) => s,
        GpuError::Library(code, msg) => {
            eprintln!("ark-circom-amd: libg16_amd failed in {site} (status {code}): {msg}");
            SynthesisError::UnexpectedIdentity
        }
    }
}

impl Groth16Gpu {
    /// `create_proof_with_reduction_and_matrices(&pk, r, s, &matrices, num_inputs, num_constraints,
    /// &full_assignment)`: `pk` and `matrices` are the ones `prover` was built from.
    pub fn create_proof_with_reduction_and_matrices(
        prover: &mut GpuProver,
        r: Fr,
        s: Fr,
        num_inputs: usize,
        num_constraints: usize,
        full_assignment: &[Fr],
    ) -> Result<Proof<Bn254>, SynthesisError> {
        Self::try_create_proof_with_reduction_and_matrices(prover, r, s, num_inputs, num_constraints, full_assignment)
            .map_err(|e| to_synthesis(e, "create_proof_with_reduction_and_matrices"))
    }

    /// the same with the typed error
    pub fn try_create_proof_with_reduction_and_matrices(
        prover: &mut GpuProver,
        r: Fr,
        s: Fr,
        num_inputs: usize,
        num_constraints: usize,
        full_assignment: &[Fr],
    ) -> Result<Proof<Bn254>, GpuError> {
        if num_inputs != prover.num_inputs() || num_constraints != prover.num_constraints() {
            return Err(GpuError::Synthesis(SynthesisError::MalformedVerifyingKey));
        }
        prover.create_proof(r, s, full_assignment)
    }

    /// `SNARK::prove(&pk, circuit, rng)`: r, s from the rng; the assignment is the one
    /// `CircomCircuit::generate_constraints` allocates (src/circom/circuit.rs:35-58): through the
    /// wire mapping when the circuit carries one, `witness[i]` when it is `None` (what
    /// `CircomBuilder::build` produces, builder.rs:84-85).  No `ConstraintSystem` is synthesised:
    /// the matrices were taken from the key file once.
    pub fn prove<R: Rng>(
        prover: &mut GpuProver,
        circuit: CircomCircuit<Fr>,
        rng: &mut R,
    ) -> Result<Proof<Bn254>, SynthesisError> {
        Self::try_prove(prover, circuit, rng).map_err(|e| to_synthesis(e, "prove"))
    }

    /// the same with the typed error
    pub fn try_prove<R: Rng>(
        prover: &mut GpuProver,
        circuit: CircomCircuit<Fr>,
        rng: &mut R,
    ) -> Result<Proof<Bn254>, GpuError> {
        let w = circuit.witness.as_ref().ok_or(SynthesisError::AssignmentMissing)?;
        let n = circuit.r1cs.num_inputs + circuit.r1cs.num_aux;
        let assignment: Vec<Fr> = match &circuit.r1cs.wire_mapping {
            Some(m) => (0..n).map(|i| w[m[i]]).collect(),
            None => w[..n].to_vec(),
        };
        let r = Fr::rand(rng);
        let s = Fr::rand(rng);
        prover.create_proof(r, s, &assignment)
    }
}
