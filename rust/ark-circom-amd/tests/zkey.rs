//! The reference's two zkey integration tests (src/zkey.rs:846-919) with the GPU prover in the
//! place of `Groth16::<Bn254, CircomReduction>`; everything else is the reference's own code path.
//! Run from the ark-circom checkout root (the tests read ./test-vectors/...).
//
// Derived from arkworks-rs/circom-compat (tests at src/zkey.rs:846-919), Copyright (c) 2021 Georgios Konstantopoulos,
// licensed MIT OR Apache-2.0; this file keeps that licence (see the crate's Cargo.toml).
use std::{collections::HashMap, fs::File};

use ark_bn254::{Bn254, Fr};
use ark_circom_amd::{read_zkey, CircomBuilder, CircomConfig, GpuProver, Groth16Gpu, WitnessCalculator};
use ark_groth16::Groth16;
use ark_std::rand::thread_rng;
use wasmer::Store;

#[tokio::test]
async fn verify_proof_with_zkey_with_r1cs() {
    let path = "./test-vectors/test.zkey";
    let mut file = File::open(path).unwrap();
    let (params, matrices) = read_zkey(&mut file).unwrap();
    let mut prover = GpuProver::new(&params, &matrices).unwrap();

    let cfg = CircomConfig::<Fr>::new("./test-vectors/mycircuit_js/mycircuit.wasm", "./test-vectors/mycircuit.r1cs").unwrap();
    let mut builder = CircomBuilder::new(cfg);
    builder.push_input("a", 3);
    builder.push_input("b", 11);

    let circom = builder.build().unwrap();
    let inputs = circom.get_public_inputs().unwrap();

    let mut rng = thread_rng();
    let proof = Groth16Gpu::prove(&mut prover, circom, &mut rng).unwrap();

    let pvk = Groth16::<Bn254>::process_vk(&params.vk).unwrap();
    let verified = Groth16::<Bn254>::verify_with_processed_vk(&pvk, &inputs, &proof).unwrap();
    assert!(verified);
}

#[tokio::test]
async fn verify_proof_with_zkey_without_r1cs() {
    let path = "./test-vectors/test.zkey";
    let mut file = File::open(path).unwrap();
    let (params, matrices) = read_zkey(&mut file).unwrap();
    let mut prover = GpuProver::new(&params, &matrices).unwrap();
    let mut store = Store::default();
    let mut wtns = WitnessCalculator::new(&mut store, "./test-vectors/mycircuit_js/mycircuit.wasm").unwrap();
    let mut inputs: HashMap<String, Vec<num_bigint::BigInt>> = HashMap::new();
    inputs.entry("a".to_string()).or_insert_with(Vec::new).push(3.into());
    inputs.entry("b".to_string()).or_insert_with(Vec::new).push(11.into());

    let mut rng = thread_rng();
    use ark_std::UniformRand;
    let num_inputs = matrices.num_instance_variables;
    let num_constraints = matrices.num_constraints;
    let r = Fr::rand(&mut rng);
    let s = Fr::rand(&mut rng);

    let full_assignment = wtns.calculate_witness_element::<Fr, _>(&mut store, inputs, false).unwrap();
    let proof = Groth16Gpu::create_proof_with_reduction_and_matrices(&mut prover, r, s, num_inputs, num_constraints, full_assignment.as_slice()).unwrap();

    // bit-identical to the CPU path on the same (pk, r, s, w)
    let cpu = Groth16::<Bn254, ark_circom_amd::CircomReduction>::create_proof_with_reduction_and_matrices(
        &params, r, s, &matrices, num_inputs, num_constraints, full_assignment.as_slice(),
    )
    .unwrap();
    assert_eq!(proof, cpu);

    let pvk = Groth16::<Bn254>::process_vk(&params.vk).unwrap();
    let inputs = &full_assignment[1..num_inputs];
    let verified = Groth16::<Bn254>::verify_with_processed_vk(&pvk, inputs, &proof).unwrap();
    assert!(verified);

    // the GPU batch verifier gives the same verdicts (right input, wrong input)
    let wrong = vec![inputs[0] + Fr::from(1u64)];
    let ok = ark_circom_amd::verify_batch(&params.vk, &[inputs.to_vec(), wrong], &[proof.clone(), proof], 0).unwrap();
    assert_eq!(ok, vec![true, false]);
}
