/*
 * poseidon_mi355x.h -- C ABI of the MI355X-native batched Poseidon permutation / duplex sponge.
 *
 * Drop-in boundary for the Poseidon hot path of arkworks-rs/sponge (ark-sponge).  The reference has
 * no FFI seam of its own (src/lib.rs:10 forbids unsafe); the seam is its generic trait surface.  Each
 * entry point below names the reference interface it replaces.  The Rust-side binding a maintainer
 * adds is shown in INTEGRATION.md.
 *
 * Element representation (everywhere in this ABI): a field element is 4 little-endian uint64_t limbs
 * holding the fully reduced Montgomery residue  x * 2^256 mod p  -- byte-identical to ark-ff's
 * Fp<MontBackend<_,4>,4>, so a Rust &[Fr] / Vec<Fr> is passed as-is (src/poseidon/mod.rs:57 `state`).
 * "Bit-exact" means limb-for-limb equality of these residues.  Inputs must be fully reduced, as ark-ff keeps them:
 * config constants are checked, batch data is not, and the result for an unreduced state or message element is
 * unspecified (the permutation kernels happen to process it modulo p; the absorb driver adds elements into the state
 * as 256-bit residues with ONE conditional subtraction, which is exact only for reduced operands).
 *
 * State order inside one sponge state is the reference's: capacity elements first, then the rate
 * elements (src/poseidon/mod.rs:128,143,159).
 *
 * All functions return PMX_OK (0) or a negative pmx_status; pmx_last_error() gives the message of the
 * calling thread's last failure.  Nothing throws or aborts across the boundary (the reference panics:
 * src/poseidon/mod.rs:196-203): every entry point that allocates, locks or spawns runs inside a catch-all that turns
 * std::bad_alloc / std::system_error / anything else into PMX_ERR_HOST.  There is NO CPU fallback: without a usable HIP device every data-path
 * call fails with PMX_ERR_HIP.
 *
 * Threading: the host-buffer entry points of one pmx_ctx serialise on a lock inside the context (they share its
 * staging buffers and streams); the *_dev entry points only enqueue on the caller's stream and may be called
 * concurrently (the absorb / squeeze ones take a short lock inside the context while they enqueue).  Distinct contexts are independent.  Every call runs with its context's device current and restores
 * the calling thread's current HIP device before it returns; a `stream` argument must belong to the context's device.
 */
#ifndef POSEIDON_MI355X_H
#define POSEIDON_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: PMX_ERR_HOST, pmx_merkle_verify_paths_dev, indices >= 2^depth fail verification, pmx_ctx_engine_info,
 *    pmx_merkle_2to1_forest[_dev]; the test hooks left this header (poseidon_mi355x_testing.h).
 * 4: the benchmark diagnostics (pmx_diag_*) left the library for libposeidon_mi355x_diag.so (poseidon_mi355x_diag.h); the host-buffer
 *    absorb / squeeze take any length (they cut a call longer than 65536 rates into pieces).
 * 5: pmx_mgpu_gather_dev (the result to one rank), pmx_mgpu_permute_gather_dev (the last step and its gather, overlapped). */
#define PMX_ABI_VERSION 5
#define PMX_LIMBS 4        /* uint64_t limbs per field element */
#define PMX_MAX_WIDTH 16   /* largest rate+capacity accepted (reference default table uses 3..9) */

typedef enum pmx_status {
    PMX_OK = 0,
    PMX_ERR_CONFIG = -1,      /* violates the asserts of PoseidonConfig::new (src/poseidon/mod.rs:196-203) or a limit of this build */
    PMX_ERR_ARG = -2,         /* null pointer / bad size / bad mode word */
    PMX_ERR_HIP = -3,         /* HIP runtime failure or no device */
    PMX_ERR_UNSUPPORTED = -4, /* width without a compiled kernel */
    PMX_ERR_RCCL = -5,        /* RCCL failure in a device group (pmx_mgpu_*) */
    PMX_ERR_HOST = -6         /* host resource failure inside the library: out of memory, a lock or a thread could not be made */
} pmx_status;

/* DuplexSpongeMode (src/lib.rs:198-210) as two words per sponge: tag + index. */
#define PMX_MODE_ABSORBING 0u /* index = next_absorb_index  in [0, rate] */
#define PMX_MODE_SQUEEZING 1u /* index = next_squeeze_index in [0, rate] */

/*
 * PoseidonConfig<F> (src/poseidon/mod.rs:23-42) plus the prime.  ark / mds are borrowed for the
 * duration of pmx_ctx_create only.
 */
typedef struct pmx_config {
    uint32_t full_rounds;            /* PoseidonConfig::full_rounds (RF/2 before the partial rounds, RF - RF/2 after, mod.rs:96-116) */
    uint32_t partial_rounds;         /* PoseidonConfig::partial_rounds */
    uint64_t alpha;                  /* PoseidonConfig::alpha, S-box exponent */
    uint32_t rate;                   /* PoseidonConfig::rate */
    uint32_t capacity;               /* PoseidonConfig::capacity */
    uint64_t modulus[PMX_LIMBS];     /* p, canonical little-endian limbs (odd, < 2^256) */
    const uint64_t *ark;             /* [full_rounds+partial_rounds][rate+capacity][4]  ark[round][i] */
    const uint64_t *mds;             /* [rate+capacity][rate+capacity][4]               mds[i][j], row-major */
} pmx_config;

typedef struct pmx_ctx pmx_ctx;

/* ---- library / error ------------------------------------------------------------------------- */
int pmx_abi_version(void);
const char *pmx_last_error(void);
/* Number of HIP devices visible (0 when none; never fails). */
int pmx_device_count(void);

/* ---- pinned host memory (optional) --------------------------------------------------------------
 * The host-buffer entry points below accept any host pointer.  When a buffer is page-locked (allocated here, or
 * registered by the caller with hipHostRegister) they switch from the runtime's pageable staging to a chunked
 * H2D / kernel / D2H pipeline on two streams: 2^20 states round-trip in 3.8 ms instead of 9-15 ms. */
int pmx_host_alloc(void **ptr, size_t bytes);
int pmx_host_free(void *ptr);

/* ---- device memory for the *_dev entry points (optional) ------------------------------------------
 * A caller that already manages HIP memory passes its own device pointers and streams.  One that does not (a Rust
 * crate without HIP bindings) gets what it needs here: allocation on a device, copies ordered on a stream (NULL = the
 * device's default stream; copies from / to pageable host memory complete before the call returns, page-locked ones
 * are asynchronous), and a stream wait. */
int pmx_device_alloc(int device, void **d_ptr, size_t bytes);
int pmx_device_free(int device, void *d_ptr);
int pmx_device_upload(int device, void *d_dst, const void *h_src, size_t bytes, void *stream);
int pmx_device_download(int device, void *h_dst, const void *d_src, size_t bytes, void *stream);
int pmx_stream_synchronize(int device, void *stream);

/* ---- parameters (host only) -------------------------------------------------------------------
 * find_poseidon_ark_and_mds (src/poseidon/traits.rs:105-146) with PoseidonGrainLFSR
 * (src/poseidon/grain_lfsr.rs:15-189): width = rate+1.  Writes Montgomery residues:
 * ark_out [(full_rounds+partial_rounds)*(rate+1)*4], mds_out [(rate+1)*(rate+1)*4]. */
int pmx_find_poseidon_ark_and_mds(const uint64_t modulus[PMX_LIMBS], uint64_t prime_bits, uint32_t rate,
                                  uint32_t full_rounds, uint32_t partial_rounds, uint32_t skip_matrices,
                                  uint64_t *ark_out, uint64_t *mds_out);

/* Montgomery constants of a modulus: inv = -p^-1 mod 2^64, r = 2^256 mod p, r2 = 2^512 mod p. */
int pmx_mont_constants(const uint64_t modulus[PMX_LIMBS], uint64_t *inv, uint64_t r[PMX_LIMBS],
                       uint64_t r2[PMX_LIMBS]);
/* canonical -> Montgomery / Montgomery -> canonical, in place over n elements (ark-ff from_bigint / into_bigint). */
int pmx_to_mont(const uint64_t modulus[PMX_LIMBS], uint64_t *elems, size_t n);
int pmx_from_mont(const uint64_t modulus[PMX_LIMBS], uint64_t *elems, size_t n);

/* ---- context ----------------------------------------------------------------------------------
 * CryptographicSponge::new's parameter clone (src/poseidon/mod.rs:219-230) happens once here: the
 * config is validated like PoseidonConfig::new (src/poseidon/mod.rs:187-213) and its constants are
 * uploaded to `device`.  */
int pmx_ctx_create(const pmx_config *cfg, int device, pmx_ctx **out);
int pmx_ctx_destroy(pmx_ctx *ctx);
/* The same context from a process-wide cache keyed by (config contents, device), reference-counted: what a binding's
 * CryptographicSponge::new should call, so that a sponge per transcript (the reference's usage, mod.rs:219-230 clones
 * ~4 KB of parameters per sponge) costs a hash of the constants, not a table derivation + upload.  Contexts whose
 * count drops to zero stay resident (a few, oldest evicted first); pmx_ctx_cache_clear frees the idle ones. */
int pmx_ctx_acquire(const pmx_config *cfg, int device, pmx_ctx **out);
int pmx_ctx_release(pmx_ctx *ctx);
int pmx_ctx_cache_clear(void);
/* width t = rate + capacity of the context's config */
int pmx_ctx_width(const pmx_ctx *ctx);

/* ---- which engine a call runs on --------------------------------------------------------------------
 * The launchers pick a kernel family by width, exponent, schedule, modulus and batch size.  This
 * reports the choice for one call, through the launchers' own conditions, so that a benchmark's instruction accounting
 * cannot drift from the kernels: op = one of PMX_OP_*, n = units of the call (states, rows, compressions of one tree
 * level, sponges), len = in_len / out_len of an absorb / squeeze call (ignored otherwise).  Host only; nothing is
 * launched.  The reference has no counterpart (one code path, src/poseidon/mod.rs:95-118). */
#define PMX_OP_PERMUTE 0
#define PMX_OP_HASH 1
#define PMX_OP_COMPRESS 2
#define PMX_OP_ABSORB 3
#define PMX_OP_SQUEEZE 4
typedef struct pmx_engine_info {
    char engine[64];     /* e.g. "RegEngine<3,5,opt,tab>", "HybridEngine<9,5,mfma,windows of 6>", "... x passes", "QuadEngine<5>" */
    int width;           /* t */
    int threads;         /* per workgroup */
    int waves_per_simd;  /* the kernel's launch bound (what its register allocation is held to) */
    int lds_bytes;       /* dynamic LDS per workgroup */
    int optimised;       /* 1: optimised round schedule (sparse partial rounds, normalised layers), 0: the reference's dense one */
    int row_tables;      /* 1: t-term matrix rows consume shifted tables (81 t + 18 multiplies), 0: element form (81 t + 81);
                          * window engines: how the history terms of the S-box inputs are formed - 1 shifted tables, 2 rows on the matrix cores */
    int lane_tables;     /* 1: identity-lane updates of the sparse layers consume shifted tables */
    int mfma_dense;      /* 1: rows of the dense layers come from the matrix cores (int8 GEMM, pmx_mfma.hpp) */
    int launches;        /* kernel launches of the call: 1, or the passes of an absorb / squeeze call on wide states (and on device-filling t = 3 calls) */
    int partial_window;  /* K > 0: the partial rounds run as windows of K S-boxes, each closed by ONE layer on the matrix cores
                            (no sparse layers on the VALU); 0: one sparse layer per partial round */
} pmx_engine_info;
int pmx_ctx_engine_info(const pmx_ctx *ctx, int op, size_t n, size_t len, pmx_engine_info *out);

/* ---- permutation ------------------------------------------------------------------------------
 * PoseidonSponge::permute (src/poseidon/mod.rs:95-118 with apply_ark :76-80, apply_s_box :63-74,
 * apply_mds :82-93) applied independently to n states, in place.  states: [n][t][4].
 * The host variant copies in, runs the kernel, copies out.  The _dev variant takes a device pointer
 * and a hipStream_t (NULL = default stream) and only enqueues.  */
int pmx_permute_batch(pmx_ctx *ctx, uint64_t *states, size_t n);
int pmx_permute_batch_dev(pmx_ctx *ctx, uint64_t *d_states, size_t n, void *stream);

/* ---- fixed-shape hash driver ------------------------------------------------------------------
 * Per row: PoseidonSponge::new; absorb(in_len native elements); squeeze_native_field_elements(out_len)
 * (src/poseidon/mod.rs:219-254, 321-341).  in: [n][in_len][4], out: [n][out_len][4].  in_len may be 0
 * (absorb of an empty input is a no-op, mod.rs:234-236). */
int pmx_hash_batch(pmx_ctx *ctx, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len, size_t n);
int pmx_hash_batch_dev(pmx_ctx *ctx, const uint64_t *d_in, size_t in_len, uint64_t *d_out, size_t out_len,
                       size_t n, void *stream);

/* ---- duplex sponge driver on explicit (state, mode) ---------------------------------------------
 * n mid-stream sponges as moved out by SpongeExt::into_state (src/lib.rs:188-195,
 * src/poseidon/mod.rs:344-367): states [n][t][4], mode_tag [n], mode_index [n]; all updated in place.
 * absorb: CryptographicSponge::absorb for in_len native elements per sponge (mod.rs:232-254, 121-150).
 * squeeze: FieldBasedCryptographicSponge::squeeze_native_field_elements(out_len) (mod.rs:321-341,
 * 153-182, including the `!= rate` test of :175).  Sponges in one call may be in different modes.
 * Widths 4..9 (and width 3 from 32769 sponges up) run a call as PASSES on the permutation engine of the width (one launch per
 * permutation a sponge of the batch can need, ceil(len / rate); a sponge is permuted exactly as often as the reference would
 * permute it).  A _dev call moves at most 65536 rates of elements per sponge, whatever the width and the batch size (PMX_ERR_ARG
 * beyond: split the call - to a duplex sponge two calls are the same as one, except that a squeeze must not be cut so that a piece of
 * exactly `rate` elements meets a sponge inside its rate, mod.rs:175).  The HOST-buffer entry points take any length, as the reference
 * does: they cut a longer call into such pieces themselves.
 * The _dev variants only enqueue on the caller's stream, with two provisos for the pass form: (1) the pass lists live in device
 * blocks the context keeps in a pool (a call takes the block its stream used last, or one whose earlier use has completed - the
 * context's own event says so -, or allocates one: calls on different streams stay independent; concurrent calls of ONE context
 * serialise while they enqueue).  An allocation (hipMalloc) may therefore happen inside the call: these two entry points must not
 * be captured into a hipGraph.  (2) A stream must outlive the driver work pending on it: synchronise it before destroying it. */
int pmx_sponge_absorb_batch(pmx_ctx *ctx, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index,
                            const uint64_t *in, size_t in_len, size_t n);
int pmx_sponge_squeeze_batch(pmx_ctx *ctx, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index,
                             uint64_t *out, size_t out_len, size_t n);
int pmx_sponge_absorb_batch_dev(pmx_ctx *ctx, uint64_t *d_states, uint32_t *d_mode_tag, uint32_t *d_mode_index,
                                const uint64_t *d_in, size_t in_len, size_t n, void *stream);
int pmx_sponge_squeeze_batch_dev(pmx_ctx *ctx, uint64_t *d_states, uint32_t *d_mode_tag, uint32_t *d_mode_index,
                                 uint64_t *d_out, size_t out_len, size_t n, void *stream);

/* ---- 2-to-1 Merkle compression ------------------------------------------------------------------
 * parent = (new; absorb([left, right]); squeeze_native(1))[0]  (needs rate >= 2), level by level.
 * leaves: [n_leaves][4], n_leaves a power of two.  nodes (may be NULL): [2*n_leaves-1][4] receives the
 * leaves, then every level, root last.  root (may be NULL): [4]. */
int pmx_merkle_2to1(pmx_ctx *ctx, const uint64_t *leaves, size_t n_leaves, uint64_t *nodes, uint64_t *root);
/* Device variant: d_nodes [2*n_leaves-1][4] must already hold the leaves in its first n_leaves rows. */
int pmx_merkle_2to1_dev(pmx_ctx *ctx, uint64_t *d_nodes, size_t n_leaves, void *stream);

/* n_trees independent 2-to-1 trees of leaves_per_tree leaves each (a power of two; n_trees is any number >= 1), advanced
 * TOGETHER level by level: the narrow top levels of one tree are latency-bound (a level of <= 16384 compressions costs one
 * permutation's dependent chain whatever its width), a level of the forest is n_trees times as wide.  Layout, level-major:
 * leaves: [n_trees][leaves_per_tree][4] (tree after tree); nodes (may be NULL): [n_trees * (2*leaves_per_tree - 1)][4] receives
 * the leaves, then level 1 of every tree (tree after tree), ..., then the n_trees roots; roots (may be NULL): [n_trees][4].
 * Tree b's node j of level l (level 0 = leaves, m = leaves_per_tree) is row  n_trees*(2m - 2m/2^l) + b*(m/2^l) + j.
 * The device variant takes d_nodes with the leaves in its first n_trees*leaves_per_tree rows and only enqueues.
 * (No counterpart in the reference: a parent is new; absorb([l, r]); squeeze_native(1) as above.) */
int pmx_merkle_2to1_forest(pmx_ctx *ctx, const uint64_t *leaves, size_t n_trees, size_t leaves_per_tree, uint64_t *nodes,
                           uint64_t *roots);
int pmx_merkle_2to1_forest_dev(pmx_ctx *ctx, uint64_t *d_nodes, size_t n_trees, size_t leaves_per_tree, void *stream);

/* Authentication paths over the node array pmx_merkle_2to1 produces ([2*n_leaves-1][4]: leaves, then every level, root
 * last).  The container itself lives upstream (ark-crypto-primitives), not in arkworks-rs/sponge; a parent is
 * (new; absorb([left, right]); squeeze_native(1))[0] as above.  depth = log2(n_leaves).
 * pmx_merkle_paths: host-only gather - paths_out [k][depth][4] receives, for each indices[i], the sibling of the leaf and
 * of each ancestor, bottom-up.
 * pmx_merkle_verify_paths: k paths at once - one upload, `depth` level steps on the device (each a batched 2-to-1
 * compression of all k running nodes), one download: ok_out[i] = 1 iff hashing leaves[i] up its path (indices[i] says
 * left / right at each level) gives `root` and indices[i] < 2^depth.
 * pmx_merkle_verify_paths_dev: the same on device-resident buffers, enqueue only; d_work is [k][12] u64 of scratch. */
int pmx_merkle_paths(const uint64_t *nodes, size_t n_leaves, const uint64_t *indices, size_t k, uint64_t *paths_out);
int pmx_merkle_verify_paths(pmx_ctx *ctx, const uint64_t *leaves, const uint64_t *indices, const uint64_t *paths, size_t depth,
                            size_t k, const uint64_t root[PMX_LIMBS], uint8_t *ok_out);
int pmx_merkle_verify_paths_dev(pmx_ctx *ctx, const uint64_t *d_leaves, const uint64_t *d_indices, const uint64_t *d_paths,
                                size_t depth, size_t k, const uint64_t *d_root, uint8_t *d_ok, uint64_t *d_work, void *stream);

/* ---- device groups: the batch sharded over the GPUs of one node -------------------------------------
 * The reference is single-threaded and has no distributed code; nothing in src/poseidon/mod.rs:62-183 couples one
 * sponge state to another, so n states are cut into `world` contiguous shards (pmx_shard_bounds), one per GPU, and
 * the permutation itself needs NO collective.  RCCL over xGMI is used for the final gather of the result shards
 * (to every rank: ncclAllGather, a group of ncclBroadcasts when n is not a multiple of world; to one rank, or piece by piece
 * behind the last step: grouped ncclSend / ncclRecv) and for the 32-byte subtree roots of the sharded Merkle reduction.
 *
 * A group is either all GPUs of ONE process (pmx_mgpu_create = ncclCommInitAll; this is what a Rust caller uses:
 * BatchPoseidon::new_multi in INTEGRATION.md) or ONE rank of a multi-process job (pmx_mgpu_create_rank =
 * ncclCommInitRank; rank 0 makes the id with pmx_mgpu_unique_id and the launcher carries it to the other ranks).
 * A group holds `n_local` devices with consecutive ranks first_rank .. first_rank + n_local - 1 of `world`.
 * Arrays indexed [local] below have n_local entries.  The *_dev calls enqueue on the group's own per-device
 * streams (pmx_mgpu_stream) and return; pmx_mgpu_synchronize waits for all of them.
 *
 * RCCL is bound when the first group is formed (dlopen of librccl.so.1 by SONAME - a process that already holds a copy
 * keeps it -, then /opt/rocm/lib/librccl.so.1).  The library reads no environment variable for this: a site's own RCCL
 * build is chosen the way any shared library is, through the loader's search path. */
#define PMX_UNIQUE_ID_BYTES 128
#define PMX_MAX_LOCAL_DEVICES 16
typedef struct pmx_mgpu pmx_mgpu;
typedef struct pmx_mgpu_info {
    int world;            /* ranks the group was created for */
    int n_local;          /* devices driven by this process */
    int first_rank;       /* rank of local device 0 */
    int width;            /* t = rate + capacity */
    int rccl_version;     /* ncclGetVersion, e.g. 22707 */
    int comm_ranks;       /* ncclCommCount of the LIVE communicator: proof that RCCL joined `world` ranks */
    int comm_first_rank;  /* ncclCommUserRank of local device 0 */
    int devices[PMX_MAX_LOCAL_DEVICES]; /* HIP device of each local slot */
} pmx_mgpu_info;

/* rank `rank` of `world` owns units [start, start + count) of n: contiguous, the first n % world shards one longer.
 * Pure host arithmetic (no device needed). */
int pmx_shard_bounds(size_t n, int world, int rank, size_t *start, size_t *count);
int pmx_mgpu_unique_id(uint8_t id[PMX_UNIQUE_ID_BYTES]);
/* devices: n_devices HIP device ordinals, or NULL for 0 .. n_devices-1 */
int pmx_mgpu_create(const pmx_config *cfg, int n_devices, const int *devices, pmx_mgpu **out);
int pmx_mgpu_create_rank(const pmx_config *cfg, int device, int rank, int world, const uint8_t id[PMX_UNIQUE_ID_BYTES],
                         pmx_mgpu **out);
int pmx_mgpu_destroy(pmx_mgpu *g);
int pmx_mgpu_get_info(const pmx_mgpu *g, pmx_mgpu_info *info);
void *pmx_mgpu_stream(const pmx_mgpu *g, int local);     /* hipStream_t of local device `local` (NULL if out of range) */
/* Its context (owned by the group).  Any other per-shard work - the hash driver, absorb, squeeze - is the single-device
 * *_dev entry point called with this context on pmx_mgpu_stream(g, local); pmx_mgpu_all_gather_dev then gathers the
 * per-row outputs with row_elems = out_len. */
pmx_ctx *pmx_mgpu_ctx(const pmx_mgpu *g, int local);
int pmx_mgpu_synchronize(pmx_mgpu *g);

/* PoseidonSponge::permute (src/poseidon/mod.rs:95-118) on n states in host memory, in place, sharded over the
 * group's devices (single-process groups): every device pipelines its own shard over PCIe, all devices at once. */
int pmx_mgpu_permute_batch(pmx_mgpu *g, uint64_t *states, size_t n);
/* pmx_hash_batch (per row: new; absorb(in_len); squeeze_native(out_len)) sharded the same way: rows [start, start+count)
 * of `in` and `out` go through device g's host path. */
int pmx_mgpu_hash_batch(pmx_mgpu *g, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len, size_t n);
/* The same on device-resident shards: d_shards[local] = [count][t][4] on that device, count from
 * pmx_shard_bounds(n_total, world, first_rank + local).  Only enqueues. */
int pmx_mgpu_permute_shards_dev(pmx_mgpu *g, uint64_t *const *d_shards, size_t n_total);
/* The final gather: d_all[local] = [n_total][row_elems][4] on every device receives all shards in rank order
 * (row_elems = t for states, 1 for digests).  RCCL; only enqueues. */
int pmx_mgpu_all_gather_dev(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total,
                            size_t row_elems);
/* The gather to ONE rank: only `root` ends with all shards (d_all of its slot; the other slots' entries are not read and may be
 * NULL).  Every other rank sends its shard over its own link to the root - 1 / world of the all-gather's bytes per link.  Grouped
 * ncclSend / ncclRecv; only enqueues. */
int pmx_mgpu_gather_dev(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, size_t row_elems,
                        int root);
/* The LAST step of a job and its gather, overlapped: pmx_mgpu_permute_shards_dev in `chunks` pieces per shard (1 .. 16), piece i's
 * transfers - to rank `root`, or to every rank when root < 0 - posted on a second stream of each slot behind piece i's kernel, so
 * the links carry piece i while pieces i + 1 ... are computed.  To the streams of pmx_mgpu_stream the call is the permutation
 * followed by the gather (they wait for the transfers at the end).  d_all as above.  Only enqueues. */
int pmx_mgpu_permute_gather_dev(pmx_mgpu *g, uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, int root, int chunks);
/* 2-to-1 Merkle tree of n_leaves = world * m leaves (both powers of two): d_nodes[local] = [2m-1][4] holds that rank's
 * m leaves in its first m rows and receives its subtree (pmx_merkle_2to1_dev); the `world` subtree roots are
 * all-gathered into d_top[local] = [2*world-1][4], which then receives the top levels, root last (world = 1: [1][4]).
 * Only enqueues. */
int pmx_mgpu_merkle_2to1_dev(pmx_mgpu *g, uint64_t *const *d_nodes, uint64_t *const *d_top, size_t n_leaves);
/* Host leaves [n_leaves][4] -> root [4] (single-process groups). */
int pmx_mgpu_merkle_2to1(pmx_mgpu *g, const uint64_t *leaves, size_t n_leaves, uint64_t *root);

/* (Test hooks of the device-group code - fault injection in the host fan-out, device groups whose slots share one GPU, a
 * named collective library - are declared in poseidon_mi355x_testing.h and exist only in libposeidon_mi355x_test.so, a
 * second build of the same objects; this library neither exports nor contains them.) */

/* (The benchmark diagnostics - the multiply-issue peak and the issue slot bench.py prices its kernels against - are NOT part of this
 * library: include/poseidon_mi355x_diag.h, libposeidon_mi355x_diag.so.) */

#ifdef __cplusplus
}
#endif
#endif /* POSEIDON_MI355X_H */
