//! Drop-in `CryptographicSponge` implementation over libposeidon_mi355x.so.
//! Place this directory at src/poseidon/mi355x/ of arkworks-rs/sponge and add `pub mod mi355x;` to src/poseidon/mod.rs.
//! NOT compiled in this repository's image (no cargo/rustc); kept in sync with INTEGRATION.md by hand.
use crate::{Absorb, CryptographicSponge, DuplexSpongeMode, FieldBasedCryptographicSponge, FieldElementSize, SpongeExt,
            poseidon::{PoseidonConfig, PoseidonSpongeState}, field_cast, squeeze_field_elements_with_sizes_default_impl};
use ark_ff::{BigInteger, PrimeField};
use std::{ffi::CStr, sync::Arc};

mod ffi;

fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(ffi::pmx_last_error()) }.to_string_lossy().into_owned();
        panic!("poseidon_mi355x: {msg} (code {rc})"); // the reference panics in the same places
    }
}

/// Device context shared by all sponges cloned from one config.
struct Ctx(*mut ffi::pmx_ctx);
unsafe impl Send for Ctx {}
impl Drop for Ctx { fn drop(&mut self) { unsafe { ffi::pmx_ctx_destroy(self.0); } } }

fn limbs<F: PrimeField>(v: &[F]) -> *const u64 { v.as_ptr() as *const u64 }        // F is [u64;4] Montgomery
fn limbs_mut<F: PrimeField>(v: &mut [F]) -> *mut u64 { v.as_mut_ptr() as *mut u64 }

fn make_ctx<F: PrimeField>(p: &PoseidonConfig<F>, device: i32) -> Arc<Ctx> {
    assert_eq!(core::mem::size_of::<F>(), 32, "4-limb Montgomery fields only");
    let ark: Vec<F> = p.ark.iter().flatten().copied().collect();
    let mds: Vec<F> = p.mds.iter().flatten().copied().collect();
    let mut modulus = [0u64; 4];
    modulus.copy_from_slice(F::MODULUS.as_ref());
    let cfg = ffi::pmx_config {
        full_rounds: p.full_rounds as u32, partial_rounds: p.partial_rounds as u32, alpha: p.alpha,
        rate: p.rate as u32, capacity: p.capacity as u32, modulus, ark: limbs(&ark), mds: limbs(&mds),
    };
    let mut h = core::ptr::null_mut();
    check(unsafe { ffi::pmx_ctx_create(&cfg, device, &mut h) });
    Arc::new(Ctx(h))
}

/// `PoseidonSponge` with the permutation on an MI355X; same public fields as `poseidon::PoseidonSponge`.
#[derive(Clone)]
pub struct Mi355xPoseidonSponge<F: PrimeField> {
    pub parameters: PoseidonConfig<F>,
    pub state: Vec<F>,
    pub mode: DuplexSpongeMode,
    ctx: Arc<Ctx>,
}

impl<F: PrimeField> Mi355xPoseidonSponge<F> {
    fn mode_words(&self) -> (u32, u32) {
        match self.mode {
            DuplexSpongeMode::Absorbing { next_absorb_index } => (ffi::PMX_MODE_ABSORBING, next_absorb_index as u32),
            DuplexSpongeMode::Squeezing { next_squeeze_index } => (ffi::PMX_MODE_SQUEEZING, next_squeeze_index as u32),
        }
    }
    fn set_mode(&mut self, tag: u32, index: u32) {
        self.mode = if tag == ffi::PMX_MODE_ABSORBING {
            DuplexSpongeMode::Absorbing { next_absorb_index: index as usize }
        } else {
            DuplexSpongeMode::Squeezing { next_squeeze_index: index as usize }
        };
    }
}

impl<F: PrimeField> CryptographicSponge for Mi355xPoseidonSponge<F> {
    type Config = PoseidonConfig<F>;

    fn new(parameters: &Self::Config) -> Self {
        Self { parameters: parameters.clone(), state: vec![F::zero(); parameters.rate + parameters.capacity],
               mode: DuplexSpongeMode::Absorbing { next_absorb_index: 0 }, ctx: make_ctx(parameters, 0) }
    }

    fn absorb(&mut self, input: &impl Absorb) {
        let elems = input.to_sponge_field_elements_as_vec::<F>();       // host-side encoding stays in Rust (src/absorb.rs)
        if elems.is_empty() { return; }
        let (mut tag, mut idx) = self.mode_words();
        check(unsafe { ffi::pmx_sponge_absorb_batch(self.ctx.0, limbs_mut(&mut self.state), &mut tag, &mut idx,
                                                    limbs(&elems), elems.len(), 1) });
        self.set_mode(tag, idx);
    }

    fn squeeze_bytes(&mut self, num_bytes: usize) -> Vec<u8> {           // semantics of src/poseidon/mod.rs:256-270
        let usable_bytes = ((F::MODULUS_BIT_SIZE - 1) / 8) as usize;
        let num_elements = (num_bytes + usable_bytes - 1) / usable_bytes;
        let mut bytes = Vec::with_capacity(usable_bytes * num_elements);
        for elem in &self.squeeze_native_field_elements(num_elements) {
            bytes.extend_from_slice(&elem.into_bigint().to_bytes_le()[..usable_bytes]);
        }
        bytes.truncate(num_bytes);
        bytes
    }

    fn squeeze_bits(&mut self, num_bits: usize) -> Vec<bool> {            // semantics of src/poseidon/mod.rs:272-286
        let usable_bits = (F::MODULUS_BIT_SIZE - 1) as usize;
        let num_elements = (num_bits + usable_bits - 1) / usable_bits;
        let mut bits = Vec::with_capacity(usable_bits * num_elements);
        for elem in &self.squeeze_native_field_elements(num_elements) {
            bits.extend_from_slice(&elem.into_bigint().to_bits_le()[..usable_bits]);
        }
        bits.truncate(num_bits);
        bits
    }

    fn squeeze_field_elements_with_sizes<F2: PrimeField>(&mut self, sizes: &[FieldElementSize]) -> Vec<F2> {
        if F::characteristic() == F2::characteristic() {
            let mut buf = Vec::with_capacity(sizes.len());
            field_cast(&self.squeeze_native_field_elements_with_sizes(sizes), &mut buf).unwrap();
            buf
        } else {
            squeeze_field_elements_with_sizes_default_impl(self, sizes)
        }
    }
    // squeeze_field_elements and fork: the trait's default bodies / the reference's body apply unchanged
}

impl<F: PrimeField> FieldBasedCryptographicSponge<F> for Mi355xPoseidonSponge<F> {
    fn squeeze_native_field_elements(&mut self, num_elements: usize) -> Vec<F> {
        let mut out = vec![F::zero(); num_elements];
        let (mut tag, mut idx) = self.mode_words();
        check(unsafe { ffi::pmx_sponge_squeeze_batch(self.ctx.0, limbs_mut(&mut self.state), &mut tag, &mut idx,
                                                     limbs_mut(&mut out), num_elements, 1) });
        self.set_mode(tag, idx);
        out
    }
}

impl<F: PrimeField> SpongeExt for Mi355xPoseidonSponge<F> {
    type State = PoseidonSpongeState<F>;
    fn from_state(state: Self::State, params: &Self::Config) -> Self {
        let mut s = Self::new(params);
        s.mode = state.mode;      // (PoseidonSpongeState's fields made pub(crate))
        s.state = state.state;
        s
    }
    fn into_state(self) -> Self::State { PoseidonSpongeState { state: self.state, mode: self.mode } }
}

/// The data-parallel entry points: n sponges / n messages / one Merkle tree per call.
pub struct BatchPoseidon<F: PrimeField> { ctx: Arc<Ctx>, t: usize, _f: core::marker::PhantomData<F> }

impl<F: PrimeField> BatchPoseidon<F> {
    pub fn new(p: &PoseidonConfig<F>, device: i32) -> Self {
        Self { ctx: make_ctx(p, device), t: p.rate + p.capacity, _f: Default::default() }
    }
    /// `permute` on every `t`-element state of `states` (len = n*t), in place.
    pub fn permute(&self, states: &mut [F]) {
        assert_eq!(states.len() % self.t, 0);
        check(unsafe { ffi::pmx_permute_batch(self.ctx.0, limbs_mut(states), states.len() / self.t) });
    }
    /// per row: `new; absorb(row); squeeze_native_field_elements(out_len)`
    pub fn hash(&self, rows: &[F], in_len: usize, out_len: usize) -> Vec<F> {
        let n = if in_len == 0 { 0 } else { rows.len() / in_len };
        let mut out = vec![F::zero(); n * out_len];
        check(unsafe { ffi::pmx_hash_batch(self.ctx.0, limbs(rows), in_len, limbs_mut(&mut out), out_len, n) });
        out
    }
    /// 2-to-1 tree over `leaves` (power of two): all nodes, leaves first, root last.
    pub fn merkle(&self, leaves: &[F]) -> Vec<F> {
        let mut nodes = vec![F::zero(); 2 * leaves.len() - 1];
        check(unsafe { ffi::pmx_merkle_2to1(self.ctx.0, limbs(leaves), leaves.len(), limbs_mut(&mut nodes), core::ptr::null_mut()) });
        nodes
    }
}
