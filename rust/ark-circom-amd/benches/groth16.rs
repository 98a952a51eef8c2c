//! The reference's bench (benches/groth16.rs:13-85) with the GPU prover: same key file, same
//! witness calculator, same verification, same `groth proof {i} {j}` bench id.
//
// Derived from arkworks-rs/circom-compat (bench at benches/groth16.rs), Copyright (c) 2021 Georgios Konstantopoulos,
// licensed MIT OR Apache-2.0; this file keeps that licence (see the crate's Cargo.toml).
use criterion::{black_box, criterion_group, criterion_main, Criterion};

use ark_bn254::{Bn254, Fr};
use ark_circom_amd::{read_zkey, GpuProver, Groth16Gpu, WitnessCalculator};
use ark_groth16::Groth16;
use ark_std::rand::thread_rng;
use wasmer::Store;

use std::{collections::HashMap, fs::File};

fn bench_groth(c: &mut Criterion, num_validators: u32, num_constraints: u32) {
    let (i, j) = (num_validators, num_constraints);
    let path = format!("./test-vectors/complex-circuit/complex-circuit-{}-{}.zkey", i, j);
    let mut file = File::open(path).unwrap();
    let (params, matrices) = read_zkey(&mut file).unwrap();
    let num_inputs = matrices.num_instance_variables;
    let num_constraints = matrices.num_constraints;
    let mut prover = GpuProver::new(&params, &matrices).unwrap(); // key + matrices resident in HBM

    let inputs = {
        let mut inputs: HashMap<String, Vec<num_bigint::BigInt>> = HashMap::new();
        inputs.entry("a".to_string()).or_insert_with(Vec::new).push(3.into());
        inputs
    };
    let mut store = Store::default();
    let mut wtns = WitnessCalculator::new(&mut store, format!("./test-vectors/complex-circuit/complex-circuit-{}-{}.wasm", i, j)).unwrap();
    let full_assignment = wtns.calculate_witness_element::<Fr, _>(&mut store, inputs, false).unwrap();

    let mut rng = thread_rng();
    use ark_std::UniformRand;
    let r = Fr::rand(&mut rng);
    let s = Fr::rand(&mut rng);

    let proof = Groth16Gpu::create_proof_with_reduction_and_matrices(&mut prover, r, s, num_inputs, num_constraints, full_assignment.as_slice()).unwrap();
    let pvk = Groth16::<Bn254>::process_vk(&params.vk).unwrap();
    let inputs = &full_assignment[1..num_inputs];
    assert!(Groth16::<Bn254>::verify_with_processed_vk(&pvk, inputs, &proof).unwrap());

    c.bench_function(&format!("groth proof {} {}", i, j), |b| {
        b.iter(|| {
            black_box(
                Groth16Gpu::create_proof_with_reduction_and_matrices(&mut prover, r, s, num_inputs, num_constraints, full_assignment.as_slice()).unwrap(),
            );
        })
    });
}

// the reference's sweep (benches/groth16.rs:87-108): 10^i variables x 10^j constraints, 3 <= i <= j <= 5,
// under the same feature name; scripts/bench_sweep.py runs the same six shapes through the C ABI
cfg_if::cfg_if! {
    if #[cfg(feature = "bench-complex-all")] {
        const MIN_NUM_VARIABLES_POWER: u32 = 3;
        const MAX_NUM_VARIABLES_POWER: u32 = 5;
        const MAX_NUM_CONSTRAINTS_POWER: u32 = 5;
        fn groth_all(c: &mut Criterion) {
            for i in MIN_NUM_VARIABLES_POWER..=MAX_NUM_VARIABLES_POWER {
                for j in i..=MAX_NUM_CONSTRAINTS_POWER {
                    bench_groth(c, 10_u32.pow(i), 10_u32.pow(j));
                }
            }
        }
        criterion_group!(benches, groth_all);
    } else {
        fn groth(c: &mut Criterion) {
            bench_groth(c, 10000, 10000);
        }
        criterion_group!(benches, groth);
    }
}

criterion_main!(benches);
