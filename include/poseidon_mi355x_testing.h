/*
 * poseidon_mi355x_testing.h -- test hooks of the device-group code.  NOT part of the product ABI (poseidon_mi355x.h),
 * not in the Rust binding; the reference has no counterpart (it has no multi-device code at all,
 * src/poseidon/mod.rs:62-183).
 *
 * These symbols exist ONLY in libposeidon_mi355x_test.so: the shipped objects with pmx_mgpu.cpp compiled -DPMX_TEST_HOOKS
 * (sponge_amd/csrc/Makefile).  libposeidon_mi355x.so, the library that ships, neither exports them nor contains the state
 * they set (tests/test_abi_and_host.py checks both export tables).  State is process-wide and atomic.  The test build also
 * reads PMX_RCCL_LIBRARY=<path> when the first group is formed and binds THAT collective library (the tests' stand-in,
 * tests/fake_rccl, when the ranks of a rehearsal are separate processes); the shipped library binds RCCL by SONAME only.
 */
#ifndef POSEIDON_MI355X_TESTING_H
#define POSEIDON_MI355X_TESTING_H

#ifdef __cplusplus
extern "C" {
#endif

/* 1 (the test build is loaded). */
int pmx_test_hooks_enabled(void);

/* The host fan-out of pmx_mgpu_permute_batch / _hash_batch fails on local slot `fail_local` (-1: off) and, with
 * no_threads != 0, runs as if no worker thread could be started (the shards then go one after the other on the calling
 * thread). */
int pmx_mgpu_test_fault(int fail_local, int no_threads);

/* allow != 0: pmx_mgpu_create accepts the same HIP device in several slots (and up to PMX_MAX_LOCAL_DEVICES slots
 * whatever the number of visible devices); `devices` must then be given explicitly.  RCCL refuses two ranks on one
 * device, so this is only useful behind the stand-in collective library of tests/fake_rccl/, which lets every
 * world > 1 branch of the device-group code run on a one-GPU box. */
int pmx_mgpu_test_shared_device(int allow);

#ifdef __cplusplus
}
#endif
#endif /* POSEIDON_MI355X_TESTING_H */
