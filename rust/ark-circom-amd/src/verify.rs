//! Batch verification on the GPU: `Groth16::<Bn254>::process_vk` + `verify_with_processed_vk`
//! (reference call sites src/zkey.rs:868-870,914-916) for many proofs under one key
//! (`g16_verify_batch`, one GPU lane per proof).  The CPU call keeps working unchanged; this is for
//! batches.
use ark_bn254::{Bn254, Fr};
use ark_groth16::{Proof, VerifyingKey};

use crate::ffi;
use crate::pack;
use crate::prover::GpuError;

/// `out[i]` = `verify_with_processed_vk(&process_vk(vk), &public_inputs[i], &proofs[i])`
pub fn verify_batch(
    vk: &VerifyingKey<Bn254>,
    public_inputs: &[Vec<Fr>],
    proofs: &[Proof<Bn254>],
    device: i32,
) -> Result<Vec<bool>, GpuError> {
    let n = proofs.len();
    let n_pub = vk.gamma_abc_g1.len() - 1;
    if public_inputs.len() != n || public_inputs.iter().any(|p| p.len() != n_pub) {
        // SynthesisError::MalformedVerifyingKey in ark-groth16's prepare_inputs
        return Err(GpuError::Synthesis(ark_relations::r1cs::SynthesisError::MalformedVerifyingKey));
    }
    let ic = pack::pack_g1_vec(&vk.gamma_abc_g1);
    let mut desc = ffi::g16_vk_desc {
        alpha_g1: [0; 64],
        beta_g2: [0; 128],
        gamma_g2: [0; 128],
        delta_g2: [0; 128],
        ic: ic.as_ptr(),
        ic_count: vk.gamma_abc_g1.len() as u32,
    };
    pack::pack_g1(&vk.alpha_g1, &mut desc.alpha_g1);
    pack::pack_g2(&vk.beta_g2, &mut desc.beta_g2);
    pack::pack_g2(&vk.gamma_g2, &mut desc.gamma_g2);
    pack::pack_g2(&vk.delta_g2, &mut desc.delta_g2);
    let mut raw = vec![0u8; ffi::G16_PROOF_BYTES * n];
    for (p, o) in proofs.iter().zip(raw.chunks_exact_mut(ffi::G16_PROOF_BYTES)) {
        pack::pack_g1(&p.a, &mut o[0..64]);
        pack::pack_g2(&p.b, &mut o[64..192]);
        pack::pack_g1(&p.c, &mut o[192..256]);
    }
    let mut pubs: Vec<u64> = Vec::with_capacity(4 * n * n_pub);
    for v in public_inputs {
        pubs.extend_from_slice(&pack::fr_vec_words(v));
    }
    let mut ok = vec![0u8; n];
    let st = unsafe { ffi::g16_verify_batch(device, &desc, raw.as_ptr(), pubs.as_ptr(), n as u32, ok.as_mut_ptr()) };
    if st != ffi::G16_OK {
        return Err(GpuError::Library(st, "g16_verify_batch failed".into()));
    }
    Ok(ok.into_iter().map(|b| b != 0).collect())
}
