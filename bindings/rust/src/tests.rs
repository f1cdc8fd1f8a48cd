//! Tests of the MI355X binding.  They run INSIDE arkworks-rs/sponge (this directory placed at src/poseidon/mi355x/,
//! `#[cfg(test)] mod tests;` is already at the end of mod.rs) on a box with an MI355X:
//!
//!     POSEIDON_MI355X_LIB_DIR=/path/to/sponge_amd LD_LIBRARY_PATH=$POSEIDON_MI355X_LIB_DIR cargo test poseidon::mi355x
//!
//! NOT compiled in this repository's image (no cargo/rustc).  What they pin, strongest first:
//!   1. `differential_*`: `Mi355xPoseidonSponge` against the crate's own `PoseidonSponge` - the reference CPU path itself -
//!      on random absorb / squeeze scripts, state and mode compared after every step.  No restatement in between.
//!   2. the reference's known-answer test (src/poseidon/mod.rs:376-399) and its two differential tests of the trait
//!      surface (src/poseidon/tests.rs:71-85, 87-117), re-stated for the new type.
//!   3. the batched entry points against N reference sponges.
use super::{BatchPoseidon, DeviceStates, Mi355xPoseidonSponge};
use crate::poseidon::{find_poseidon_ark_and_mds, PoseidonConfig, PoseidonDefaultConfigField, PoseidonSponge};
use crate::test::Fr;
use crate::{absorb, CryptographicSponge, DuplexSpongeMode, FieldBasedCryptographicSponge, SpongeExt};
use ark_ff::{MontFp, PrimeField, UniformRand};
use ark_std::{rand::Rng, test_rng};

/// BASELINE's benchmark config: the (255, t = 3, 8 + 31) constants of the default table with the S-box exponent 5.
fn config_t3_alpha5() -> PoseidonConfig<Fr> {
    let (ark, mds) = find_poseidon_ark_and_mds::<Fr>(Fr::MODULUS_BIT_SIZE as u64, 2, 8, 31, 0);
    PoseidonConfig::new(8, 31, 5, mds, ark, 2, 1)
}

fn same_mode(a: &DuplexSpongeMode, b: &DuplexSpongeMode) -> bool {
    match (a, b) {
        (DuplexSpongeMode::Absorbing { next_absorb_index: x }, DuplexSpongeMode::Absorbing { next_absorb_index: y }) => x == y,
        (DuplexSpongeMode::Squeezing { next_squeeze_index: x }, DuplexSpongeMode::Squeezing { next_squeeze_index: y }) => x == y,
        _ => false,
    }
}

/// The reference's only permutation-output KAT (src/poseidon/mod.rs:376-399), through the new type.
#[test]
fn known_answer_of_the_reference_sponge() {
    let params = Fr::get_default_poseidon_parameters(2, false).unwrap();
    let mut sponge = Mi355xPoseidonSponge::<Fr>::new(&params);
    sponge.absorb(&vec![Fr::from(0u8), Fr::from(1u8), Fr::from(2u8)]);
    let out = sponge.squeeze_native_field_elements(3);
    let want: [Fr; 3] = [
        MontFp!("40442793463571304028337753002242186710310163897048962278675457993207843616876"),
        MontFp!("2664374461699898000291153145224099287711224021716202960480903840045233645301"),
        MontFp!("50191078828066923662070228256530692951801504043422844038937334196346054068797"),
    ];
    assert_eq!(out, want.to_vec());
}

/// One random script of absorbs and squeezes on both types, everything compared after every step.
fn run_script(params: &PoseidonConfig<Fr>, steps: usize, seed_skips: usize) {
    let mut rng = test_rng();
    for _ in 0..seed_skips { let _ = Fr::rand(&mut rng); }
    let mut cpu = PoseidonSponge::<Fr>::new(params);
    let mut gpu = Mi355xPoseidonSponge::<Fr>::new(params);
    let rate = params.rate;
    for step in 0..steps {
        // lengths around the interesting points: nothing, one, the rate exactly (lazy permutation; src/poseidon/mod.rs:175 on
        // the squeeze side), just over it, several rates
        let len = match rng.gen_range(0..6) { 0 => 0, 1 => 1, 2 => rate, 3 => rate + 1, 4 => 2 * rate + 1, _ => rng.gen_range(0..3 * rate + 2) };
        if rng.gen_bool(0.5) {
            let elems: Vec<Fr> = (0..len).map(|_| Fr::rand(&mut rng)).collect();
            cpu.absorb(&elems);
            gpu.absorb(&elems);
        } else {
            assert_eq!(cpu.squeeze_native_field_elements(len), gpu.squeeze_native_field_elements(len), "step {step}: squeeze({len})");
        }
        assert_eq!(cpu.state, gpu.state, "step {step}: state");
        assert!(same_mode(&cpu.mode, &gpu.mode), "step {step}: mode");
    }
    // and the byte / bit / non-native squeezes on the final state
    let (mut a, mut b) = (cpu.clone(), gpu.clone());
    assert_eq!(a.squeeze_bytes(100), b.squeeze_bytes(100));
    assert_eq!(cpu.squeeze_bits(600), gpu.squeeze_bits(600));
}

#[test]
fn differential_against_the_reference_sponge_default_table() {
    for rate in 2..=8 {                                   // every width of the default table: t = 3 .. 9
        for weights in [false, true] {
            run_script(&Fr::get_default_poseidon_parameters(rate, weights).unwrap(), 40, rate + weights as usize);
        }
    }
}

#[test]
fn differential_against_the_reference_sponge_benchmark_config() {
    run_script(&config_t3_alpha5(), 200, 7);
}

/// src/poseidon/tests.rs:71-85: squeezing native elements and squeezing "field elements of the same field" agree.
#[test]
fn native_cast_equals_native_squeeze() {
    let params = Fr::get_default_poseidon_parameters(2, false).unwrap();
    let mut rng = test_rng();
    let mut first = Mi355xPoseidonSponge::<Fr>::new(&params);
    first.absorb(&Fr::rand(&mut rng));
    let mut second = first.clone();
    assert_eq!(first.squeeze_native_field_elements(5), second.squeeze_field_elements::<Fr>(5));
}

/// src/poseidon/tests.rs:87-100: `absorb!` of several inputs equals absorbing them one after the other.
#[test]
fn absorb_macro_equals_sequential_absorbs() {
    let params = Fr::get_default_poseidon_parameters(2, false).unwrap();
    let mut one_by_one = Mi355xPoseidonSponge::<Fr>::new(&params);
    one_by_one.absorb(&vec![1, 2, 3, 4, 5, 6]);
    one_by_one.absorb(&Fr::from(114514u128));
    let mut at_once = Mi355xPoseidonSponge::<Fr>::new(&params);
    absorb!(&mut at_once, vec![1, 2, 3, 4, 5, 6], Fr::from(114514u128));
    assert_eq!(at_once.squeeze_native_field_elements(3), one_by_one.squeeze_native_field_elements(3));
}

/// SpongeExt (src/lib.rs:188-195): a state moved out of a reference sponge continues identically on the device, and back.
#[test]
fn state_moves_between_the_two_types() {
    let params = config_t3_alpha5();
    let mut rng = test_rng();
    let mut cpu = PoseidonSponge::<Fr>::new(&params);
    cpu.absorb(&(0..5).map(|_| Fr::rand(&mut rng)).collect::<Vec<_>>());
    let _ = cpu.squeeze_native_field_elements(1);
    let mut gpu = Mi355xPoseidonSponge::<Fr>::from_state(cpu.clone().into_state(), &params);
    assert_eq!(cpu.squeeze_native_field_elements(4), gpu.squeeze_native_field_elements(4));
    let mut back = PoseidonSponge::<Fr>::from_state(gpu.clone().into_state(), &params);
    assert_eq!(back.squeeze_native_field_elements(3), gpu.squeeze_native_field_elements(3));
}

/// The batched entry points against N independent reference sponges (src/poseidon/mod.rs:62-183 has no cross-state flow).
#[test]
fn batch_entry_points_equal_n_reference_sponges() {
    let params = config_t3_alpha5();
    let (t, n) = (3usize, 1000usize);
    let mut rng = test_rng();
    let batch = BatchPoseidon::<Fr>::new(&params, 0);
    // hash: new; absorb(4); squeeze(2) per row
    let rows: Vec<Fr> = (0..n * 4).map(|_| Fr::rand(&mut rng)).collect();
    let got = batch.hash(&rows, 4, 2);
    for (i, row) in rows.chunks(4).enumerate() {
        let mut s = PoseidonSponge::<Fr>::new(&params);
        s.absorb(&row.to_vec());
        assert_eq!(&got[2 * i..2 * i + 2], &s.squeeze_native_field_elements(2)[..], "row {i}");
    }
    // permute: a reference sponge holding the same state in mode Squeezing{rate} permutes before it emits the rate portion
    let mut states: Vec<Fr> = (0..n * t).map(|_| Fr::rand(&mut rng)).collect();
    let before = states.clone();
    batch.permute(&mut states);
    for i in (0..n).step_by(97) {
        let mut s = PoseidonSponge::<Fr>::new(&params);
        s.state = before[t * i..t * i + t].to_vec();
        s.mode = DuplexSpongeMode::Squeezing { next_squeeze_index: params.rate };   // the next squeeze permutes first
        let out = s.squeeze_native_field_elements(params.rate);
        assert_eq!(&states[t * i + params.capacity..t * i + t], &out[..], "state {i}");
        assert_eq!(&states[t * i..t * i + t], &s.state[..], "state {i} (whole)");
    }
    // the same batch kept in HBM
    let mut resident = DeviceStates::<Fr>::upload(&params, 0, &before);
    resident.permute();
    assert_eq!(resident.download(), states);
    // 2-to-1 tree: a parent is new; absorb([l, r]); squeeze(1)
    let leaves: Vec<Fr> = (0..64).map(|_| Fr::rand(&mut rng)).collect();
    let nodes = batch.merkle(&leaves);
    let mut level: Vec<Fr> = leaves.clone();
    let mut at = 64;
    while level.len() > 1 {
        let parents: Vec<Fr> = level.chunks(2).map(|p| { let mut s = PoseidonSponge::<Fr>::new(&params); s.absorb(&p.to_vec()); s.squeeze_native_field_elements(1)[0] }).collect();
        assert_eq!(&nodes[at..at + parents.len()], &parents[..]);
        at += parents.len();
        level = parents;
    }
    assert_eq!(batch.merkle_root(&leaves), *nodes.last().unwrap());
    // a forest: four trees of 16 leaves over the same 64 leaves = the level of the one tree that has four nodes
    assert_eq!(batch.merkle_forest_roots(&leaves, 4), nodes[2 * 64 - 8..2 * 64 - 4].to_vec());
    let idx = [0u64, 5, 63];
    let paths = batch.merkle_paths(&nodes, &idx);
    let picked: Vec<Fr> = idx.iter().map(|&i| leaves[i as usize]).collect();
    assert_eq!(batch.verify_paths(&picked, &idx, &paths, nodes.last().unwrap()), vec![true, true, true]);
}
