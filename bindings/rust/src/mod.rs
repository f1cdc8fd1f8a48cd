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

/// Device context of one (config contents, device), taken from the library's process-wide cache: every sponge made
/// from an equal config - however many times `new` is called - shares one set of device tables (pmx_ctx_acquire).
/// The host-buffer entry points serialise on a lock inside the context, so sharing it across threads is sound.
struct Ctx(*mut ffi::pmx_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx { fn drop(&mut self) { unsafe { ffi::pmx_ctx_release(self.0); } } }

fn limbs<F: PrimeField>(v: &[F]) -> *const u64 { v.as_ptr() as *const u64 }        // F is [u64;4] Montgomery
fn limbs_mut<F: PrimeField>(v: &mut [F]) -> *mut u64 { v.as_mut_ptr() as *mut u64 }

/// Flattens a `PoseidonConfig` into the C view and hands it to `f` (the pointers are borrowed for the call only).
fn with_c_config<F: PrimeField, R>(p: &PoseidonConfig<F>, f: impl FnOnce(&ffi::pmx_config) -> R) -> R {
    assert_eq!(core::mem::size_of::<F>(), 32, "4-limb Montgomery fields only");
    let ark: Vec<F> = p.ark.iter().flatten().copied().collect();
    let mds: Vec<F> = p.mds.iter().flatten().copied().collect();
    let mut modulus = [0u64; 4];
    modulus.copy_from_slice(F::MODULUS.as_ref());
    f(&ffi::pmx_config {
        full_rounds: p.full_rounds as u32, partial_rounds: p.partial_rounds as u32, alpha: p.alpha,
        rate: p.rate as u32, capacity: p.capacity as u32, modulus, ark: limbs(&ark), mds: limbs(&mds),
    })
}

/// `CryptographicSponge::new` calls this once per sponge (the reference clones ~4 KB of parameters there,
/// src/poseidon/mod.rs:219-230): the cost here is flattening + hashing the constants; the table derivation, the
/// device allocation and the upload happen once per distinct config and device.
fn make_ctx<F: PrimeField>(p: &PoseidonConfig<F>, device: i32) -> Arc<Ctx> {
    with_c_config(p, |cfg| {
        let mut h = core::ptr::null_mut();
        check(unsafe { ffi::pmx_ctx_acquire(cfg, device, &mut h) });
        Arc::new(Ctx(h))
    })
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

/// All GPUs of this process as one device group (pmx_mgpu_create = ncclCommInitAll): contiguous shards, no collective on
/// the permutation, RCCL for the gather of device-resident results and for the subtree roots of the sharded tree.
struct Group(*mut ffi::pmx_mgpu);
unsafe impl Send for Group {}
impl Drop for Group { fn drop(&mut self) { unsafe { ffi::pmx_mgpu_destroy(self.0); } } }

/// The data-parallel entry points: n sponges / n messages / one Merkle tree per call.
pub struct BatchPoseidon<F: PrimeField> { ctx: Arc<Ctx>, group: Option<Group>, t: usize, _f: core::marker::PhantomData<F> }

impl<F: PrimeField> BatchPoseidon<F> {
    pub fn new(p: &PoseidonConfig<F>, device: i32) -> Self {
        Self { ctx: make_ctx(p, device), group: None, t: p.rate + p.capacity, _f: Default::default() }
    }
    /// The same over `n_devices` GPUs of this node (0 = all visible ones): `permute` and `merkle_root` shard their input.
    pub fn new_multi(p: &PoseidonConfig<F>, n_devices: i32) -> Self {
        let n = if n_devices == 0 { unsafe { ffi::pmx_device_count() } } else { n_devices };
        let group = with_c_config(p, |cfg| {
            let mut g = core::ptr::null_mut();
            check(unsafe { ffi::pmx_mgpu_create(cfg, n, core::ptr::null(), &mut g) });
            Group(g)
        });
        Self { ctx: make_ctx(p, 0), group: Some(group), t: p.rate + p.capacity, _f: Default::default() }
    }
    /// Ranks RCCL actually joined (1 without a group).
    pub fn ranks(&self) -> usize {
        match &self.group {
            None => 1,
            Some(g) => {
                let mut info = core::mem::MaybeUninit::<ffi::pmx_mgpu_info>::zeroed();
                check(unsafe { ffi::pmx_mgpu_get_info(g.0, info.as_mut_ptr()) });
                unsafe { info.assume_init() }.comm_ranks as usize
            }
        }
    }
    /// `permute` on every `t`-element state of `states` (len = n*t), in place; sharded over the group's devices if any.
    pub fn permute(&self, states: &mut [F]) {
        assert_eq!(states.len() % self.t, 0);
        let n = states.len() / self.t;
        match &self.group {
            Some(g) => check(unsafe { ffi::pmx_mgpu_permute_batch(g.0, limbs_mut(states), n) }),
            None => check(unsafe { ffi::pmx_permute_batch(self.ctx.0, limbs_mut(states), n) }),
        }
    }
    /// Root of the 2-to-1 tree over `leaves` (power of two): one subtree per device, roots gathered over RCCL.
    pub fn merkle_root(&self, leaves: &[F]) -> F {
        let mut root = [F::zero()];
        match &self.group {
            Some(g) => check(unsafe { ffi::pmx_mgpu_merkle_2to1(g.0, limbs(leaves), leaves.len(), limbs_mut(&mut root)) }),
            None => check(unsafe { ffi::pmx_merkle_2to1(self.ctx.0, limbs(leaves), leaves.len(), core::ptr::null_mut(), limbs_mut(&mut root)) }),
        }
        root[0]
    }
    /// per row: `new; absorb(row); squeeze_native_field_elements(out_len)`
    pub fn hash(&self, rows: &[F], in_len: usize, out_len: usize) -> Vec<F> {
        let n = if in_len == 0 { 0 } else { rows.len() / in_len };
        let mut out = vec![F::zero(); n * out_len];
        match &self.group {
            Some(g) => check(unsafe { ffi::pmx_mgpu_hash_batch(g.0, limbs(rows), in_len, limbs_mut(&mut out), out_len, n) }),
            None => check(unsafe { ffi::pmx_hash_batch(self.ctx.0, limbs(rows), in_len, limbs_mut(&mut out), out_len, n) }),
        }
        out
    }
    /// Authentication paths of `indices` over a node array made by `merkle`: `[k][depth]` siblings, bottom-up.
    pub fn merkle_paths(&self, nodes: &[F], indices: &[u64]) -> Vec<F> {
        let n_leaves = (nodes.len() + 1) / 2;
        assert!(n_leaves.is_power_of_two() && nodes.len() == 2 * n_leaves - 1, "nodes is not a node array made by merkle()");
        let depth = n_leaves.trailing_zeros() as usize;
        let mut paths = vec![F::zero(); indices.len() * depth];
        check(unsafe { ffi::pmx_merkle_paths(limbs(nodes), n_leaves, indices.as_ptr(), indices.len(), limbs_mut(&mut paths)) });
        paths
    }
    /// `k` paths checked at once (one upload, one device step per level, one download): `true` where `leaves[i]` hashes up
    /// to `root` along `paths[i]` and `indices[i]` names a leaf of a tree of that depth.
    pub fn verify_paths(&self, leaves: &[F], indices: &[u64], paths: &[F], root: &F) -> Vec<bool> {
        let k = leaves.len();
        assert_eq!(indices.len(), k, "one index per leaf");
        assert!(k == 0 || paths.len() % k == 0, "paths is not [k][depth]");
        let depth = if k == 0 { 0 } else { paths.len() / k };
        let mut ok = vec![0u8; k];
        check(unsafe { ffi::pmx_merkle_verify_paths(self.ctx.0, limbs(leaves), indices.as_ptr(), limbs(paths), depth, k,
                                                    limbs(core::slice::from_ref(root)), ok.as_mut_ptr()) });
        ok.into_iter().map(|b| b != 0).collect()
    }
    /// Roots of `n_trees` trees over `leaves` (tree after tree, each a power of two long), advanced together level by level:
    /// the narrow top levels of a single tree are latency-bound, a level of the forest is `n_trees` times as wide.
    pub fn merkle_forest_roots(&self, leaves: &[F], n_trees: usize) -> Vec<F> {
        assert!(n_trees > 0 && leaves.len() % n_trees == 0, "leaves is not [n_trees][leaves_per_tree]");
        let mut roots = vec![F::zero(); n_trees];
        check(unsafe { ffi::pmx_merkle_2to1_forest(self.ctx.0, limbs(leaves), n_trees, leaves.len() / n_trees, core::ptr::null_mut(),
                                                   limbs_mut(&mut roots)) });
        roots
    }
    /// 2-to-1 tree over `leaves` (power of two): all nodes, leaves first, root last.
    pub fn merkle(&self, leaves: &[F]) -> Vec<F> {
        let mut nodes = vec![F::zero(); 2 * leaves.len() - 1];
        check(unsafe { ffi::pmx_merkle_2to1(self.ctx.0, limbs(leaves), leaves.len(), limbs_mut(&mut nodes), core::ptr::null_mut()) });
        nodes
    }
}

/// A batch that stays in HBM between calls - the `_dev` entry points without any HIP binding on the Rust side
/// (`pmx_device_alloc` / `_upload` / `_download`): upload once, permute as often as needed, download once.
pub struct DeviceStates<F: PrimeField> {
    ptr: *mut core::ffi::c_void,
    n: usize,
    t: usize,
    device: i32,
    ctx: Arc<Ctx>,
    _f: core::marker::PhantomData<F>,
}

impl<F: PrimeField> DeviceStates<F> {
    pub fn upload(p: &PoseidonConfig<F>, device: i32, states: &[F]) -> Self {
        let t = p.rate + p.capacity;
        assert_eq!(states.len() % t, 0);
        let mut ptr = core::ptr::null_mut();
        check(unsafe { ffi::pmx_device_alloc(device, &mut ptr, states.len() * 32) });
        check(unsafe { ffi::pmx_device_upload(device, ptr, states.as_ptr() as *const _, states.len() * 32, core::ptr::null_mut()) });
        Self { ptr, n: states.len() / t, t, device, ctx: make_ctx(p, device), _f: Default::default() }
    }
    /// `permute` on every state, in place in device memory (enqueued on the device's default stream).
    pub fn permute(&mut self) {
        check(unsafe { ffi::pmx_permute_batch_dev(self.ctx.0, self.ptr as *mut u64, self.n, core::ptr::null_mut()) });
    }
    pub fn download(&self) -> Vec<F> {
        let mut out = vec![F::zero(); self.n * self.t];
        check(unsafe { ffi::pmx_device_download(self.device, out.as_mut_ptr() as *mut _, self.ptr, out.len() * 32, core::ptr::null_mut()) });
        check(unsafe { ffi::pmx_stream_synchronize(self.device, core::ptr::null_mut()) });
        out
    }
}

impl<F: PrimeField> Drop for DeviceStates<F> {
    fn drop(&mut self) { unsafe { ffi::pmx_device_free(self.device, self.ptr); } }
}

#[cfg(test)]
mod tests;   // needs an MI355X: the reference's KATs and a differential test against the crate's own PoseidonSponge
