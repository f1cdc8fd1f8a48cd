//! Raw bindings of include/poseidon_mi355x.h.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct pmx_config {
    pub full_rounds: u32,
    pub partial_rounds: u32,
    pub alpha: u64,
    pub rate: u32,
    pub capacity: u32,
    pub modulus: [u64; 4],
    pub ark: *const u64, // [full_rounds+partial_rounds][rate+capacity][4]
    pub mds: *const u64, // [rate+capacity][rate+capacity][4], mds[i][j] row-major
}
#[repr(C)]
pub struct pmx_ctx { _private: [u8; 0] }
#[repr(C)]
pub struct pmx_mgpu { _private: [u8; 0] }
pub const PMX_UNIQUE_ID_BYTES: usize = 128;
pub const PMX_MAX_LOCAL_DEVICES: usize = 16;
#[repr(C)]
pub struct pmx_mgpu_info {
    pub world: c_int,
    pub n_local: c_int,
    pub first_rank: c_int,
    pub width: c_int,
    pub rccl_version: c_int,
    pub comm_ranks: c_int,
    pub comm_first_rank: c_int,
    pub devices: [c_int; PMX_MAX_LOCAL_DEVICES],
}
#[repr(C)]
pub struct pmx_engine_info {
    pub engine: [c_char; 64],
    pub width: c_int,
    pub threads: c_int,
    pub waves_per_simd: c_int,
    pub lds_bytes: c_int,
    pub optimised: c_int,
    pub row_tables: c_int,
    pub lane_tables: c_int,
    pub mfma_dense: c_int,
    pub launches: c_int,
    pub partial_window: c_int,
}
pub const PMX_OP_PERMUTE: c_int = 0;
pub const PMX_OP_HASH: c_int = 1;
pub const PMX_OP_COMPRESS: c_int = 2;
pub const PMX_OP_ABSORB: c_int = 3;
pub const PMX_OP_SQUEEZE: c_int = 4;

pub const PMX_MODE_ABSORBING: u32 = 0;
pub const PMX_MODE_SQUEEZING: u32 = 1;

extern "C" {
    pub fn pmx_last_error() -> *const c_char;
    pub fn pmx_ctx_create(cfg: *const pmx_config, device: c_int, out: *mut *mut pmx_ctx) -> c_int;
    pub fn pmx_ctx_destroy(ctx: *mut pmx_ctx) -> c_int;
    pub fn pmx_host_alloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn pmx_host_free(ptr: *mut c_void) -> c_int;
    pub fn pmx_permute_batch(ctx: *mut pmx_ctx, states: *mut u64, n: usize) -> c_int;
    pub fn pmx_permute_batch_dev(ctx: *mut pmx_ctx, d_states: *mut u64, n: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_hash_batch(ctx: *mut pmx_ctx, input: *const u64, in_len: usize, out: *mut u64, out_len: usize, n: usize) -> c_int;
    pub fn pmx_sponge_absorb_batch(ctx: *mut pmx_ctx, states: *mut u64, mode_tag: *mut u32, mode_index: *mut u32,
                                   input: *const u64, in_len: usize, n: usize) -> c_int;
    pub fn pmx_sponge_squeeze_batch(ctx: *mut pmx_ctx, states: *mut u64, mode_tag: *mut u32, mode_index: *mut u32,
                                    out: *mut u64, out_len: usize, n: usize) -> c_int;
    pub fn pmx_merkle_2to1(ctx: *mut pmx_ctx, leaves: *const u64, n_leaves: usize, nodes: *mut u64, root: *mut u64) -> c_int;
    pub fn pmx_find_poseidon_ark_and_mds(modulus: *const u64, prime_bits: u64, rate: u32, full_rounds: u32,
                                         partial_rounds: u32, skip_matrices: u32, ark_out: *mut u64, mds_out: *mut u64) -> c_int;
    // the rest of the header: not needed by the trait shim in mod.rs, declared so that the binding is complete
    pub fn pmx_abi_version() -> c_int;
    pub fn pmx_device_count() -> c_int;
    pub fn pmx_ctx_width(ctx: *const pmx_ctx) -> c_int;
    pub fn pmx_ctx_engine_info(ctx: *const pmx_ctx, op: c_int, n: usize, len: usize, out: *mut pmx_engine_info) -> c_int;
    pub fn pmx_mont_constants(modulus: *const u64, inv: *mut u64, r: *mut u64, r2: *mut u64) -> c_int;
    pub fn pmx_to_mont(modulus: *const u64, elems: *mut u64, n: usize) -> c_int;
    pub fn pmx_from_mont(modulus: *const u64, elems: *mut u64, n: usize) -> c_int;
    pub fn pmx_hash_batch_dev(ctx: *mut pmx_ctx, d_in: *const u64, in_len: usize, d_out: *mut u64, out_len: usize, n: usize,
                              stream: *mut c_void) -> c_int;
    pub fn pmx_sponge_absorb_batch_dev(ctx: *mut pmx_ctx, d_states: *mut u64, d_mode_tag: *mut u32, d_mode_index: *mut u32,
                                       d_in: *const u64, in_len: usize, n: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_sponge_squeeze_batch_dev(ctx: *mut pmx_ctx, d_states: *mut u64, d_mode_tag: *mut u32, d_mode_index: *mut u32,
                                        d_out: *mut u64, out_len: usize, n: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_merkle_2to1_dev(ctx: *mut pmx_ctx, d_nodes: *mut u64, n_leaves: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_merkle_2to1_forest(ctx: *mut pmx_ctx, leaves: *const u64, n_trees: usize, leaves_per_tree: usize, nodes: *mut u64,
                                  roots: *mut u64) -> c_int;
    pub fn pmx_merkle_2to1_forest_dev(ctx: *mut pmx_ctx, d_nodes: *mut u64, n_trees: usize, leaves_per_tree: usize,
                                      stream: *mut c_void) -> c_int;
    pub fn pmx_merkle_paths(nodes: *const u64, n_leaves: usize, indices: *const u64, k: usize, paths_out: *mut u64) -> c_int;
    pub fn pmx_merkle_verify_paths(ctx: *mut pmx_ctx, leaves: *const u64, indices: *const u64, paths: *const u64, depth: usize,
                                   k: usize, root: *const u64, ok_out: *mut u8) -> c_int;
    pub fn pmx_merkle_verify_paths_dev(ctx: *mut pmx_ctx, d_leaves: *const u64, d_indices: *const u64, d_paths: *const u64,
                                       depth: usize, k: usize, d_root: *const u64, d_ok: *mut u8, d_work: *mut u64,
                                       stream: *mut c_void) -> c_int;
    // device memory for the *_dev entry points
    pub fn pmx_device_alloc(device: c_int, d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn pmx_device_free(device: c_int, d_ptr: *mut c_void) -> c_int;
    pub fn pmx_device_upload(device: c_int, d_dst: *mut c_void, h_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_device_download(device: c_int, h_dst: *mut c_void, d_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn pmx_stream_synchronize(device: c_int, stream: *mut c_void) -> c_int;
    // shared contexts
    pub fn pmx_ctx_acquire(cfg: *const pmx_config, device: c_int, out: *mut *mut pmx_ctx) -> c_int;
    pub fn pmx_ctx_release(ctx: *mut pmx_ctx) -> c_int;
    pub fn pmx_ctx_cache_clear() -> c_int;
    // device groups
    pub fn pmx_shard_bounds(n: usize, world: c_int, rank: c_int, start: *mut usize, count: *mut usize) -> c_int;
    pub fn pmx_mgpu_unique_id(id: *mut u8) -> c_int;
    pub fn pmx_mgpu_create(cfg: *const pmx_config, n_devices: c_int, devices: *const c_int, out: *mut *mut pmx_mgpu) -> c_int;
    pub fn pmx_mgpu_create_rank(cfg: *const pmx_config, device: c_int, rank: c_int, world: c_int, id: *const u8,
                                out: *mut *mut pmx_mgpu) -> c_int;
    pub fn pmx_mgpu_destroy(g: *mut pmx_mgpu) -> c_int;
    pub fn pmx_mgpu_get_info(g: *const pmx_mgpu, info: *mut pmx_mgpu_info) -> c_int;
    pub fn pmx_mgpu_stream(g: *const pmx_mgpu, local: c_int) -> *mut c_void;
    pub fn pmx_mgpu_ctx(g: *const pmx_mgpu, local: c_int) -> *mut pmx_ctx;
    pub fn pmx_mgpu_synchronize(g: *mut pmx_mgpu) -> c_int;
    pub fn pmx_mgpu_permute_batch(g: *mut pmx_mgpu, states: *mut u64, n: usize) -> c_int;
    pub fn pmx_mgpu_hash_batch(g: *mut pmx_mgpu, input: *const u64, in_len: usize, out: *mut u64, out_len: usize, n: usize) -> c_int;
    pub fn pmx_mgpu_permute_shards_dev(g: *mut pmx_mgpu, d_shards: *const *mut u64, n_total: usize) -> c_int;
    pub fn pmx_mgpu_all_gather_dev(g: *mut pmx_mgpu, d_shards: *const *const u64, d_all: *const *mut u64, n_total: usize,
                                   row_elems: usize) -> c_int;
    pub fn pmx_mgpu_gather_dev(g: *mut pmx_mgpu, d_shards: *const *const u64, d_all: *const *mut u64, n_total: usize, row_elems: usize,
                               root: c_int) -> c_int;
    pub fn pmx_mgpu_permute_gather_dev(g: *mut pmx_mgpu, d_shards: *const *mut u64, d_all: *const *mut u64, n_total: usize, root: c_int,
                                       chunks: c_int) -> c_int;
    pub fn pmx_mgpu_merkle_2to1_dev(g: *mut pmx_mgpu, d_nodes: *const *mut u64, d_top: *const *mut u64, n_leaves: usize) -> c_int;
    pub fn pmx_mgpu_merkle_2to1(g: *mut pmx_mgpu, leaves: *const u64, n_leaves: usize, root: *mut u64) -> c_int;
}
