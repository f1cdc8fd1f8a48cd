/* Benchmark diagnostics of the MI355X Poseidon kernels: libposeidon_mi355x_diag.so (sponge_amd/csrc/pmx_diag.hip), a library of its own -
 * a caller of the sponge never needs it, bench.py does.  Nothing here touches a pmx_ctx; errors come back as the status codes of
 * poseidon_mi355x.h (PMX_ERR_ARG / PMX_ERR_HIP / PMX_ERR_HOST) with a message in pmx_diag_last_error(). */
#ifndef POSEIDON_MI355X_DIAG_H
#define POSEIDON_MI355X_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* message of the last failing pmx_diag_* call on this thread ("" if none) */
const char *pmx_diag_last_error(void);

/* ---- the multiply-issue peak -------------------------------------------------------------------------------
 * The binding roofline of these kernels is the issue rate of v_mad_u64_u32 (one per 32x32-bit limb product), not
 * HBM.  It depends on the clock the chip holds under load, so it is measured, per device and per run: a dense loop
 * of that instruction on every SIMD for about `seconds` (default 0.02).  lane_mads_per_s = median of the later
 * launches; shader_clock_hz from s_memtime / s_memrealtime inside the kernel; theoretical = CUs x 4 SIMDs x 16
 * lanes per clock (a half-rate instruction) x that clock. */
typedef struct pmx_valu_peak {
    double lane_mads_per_s;             /* the faster of the two forms below */
    double lane_mads_per_s_vcc;         /* carry-out of every multiply written to VCC */
    double lane_mads_per_s_sgpr;        /* ... to an allocator-chosen SGPR pair (what compiled kernels do) */
    double best_lane_mads_per_s;
    double shader_clock_hz;
    double theoretical_lane_mads_per_s;
    int compute_units;
    int launches;
} pmx_valu_peak;
int pmx_diag_int_valu_peak(int device, double seconds, pmx_valu_peak *out);

/* The VALU issue slot: nanoseconds per VALU instruction and SIMD of three calibration streams (12 multiplies + 4 simple
 * instructions - the permutation kernels' own mix -, 4 + 12, multiplies only) with exactly `waves_per_simd` waves resident
 * on every SIMD, measured on this device for about `seconds` (default 0.03).  A kernel's issue floor is its VALU instruction
 * count x ns_floor (the fastest of the three: no stream that holds multiplies was seen to issue faster in this run). */
typedef struct pmx_issue_slot {
    double ns_12mad_4simple;
    double ns_4mad_12simple;
    double ns_16mad;
    double ns_floor;
    int waves_per_simd;
    int compute_units;
    int launches;
} pmx_issue_slot;
int pmx_diag_issue_slot(int device, int waves_per_simd, double seconds, pmx_issue_slot *out);

#ifdef __cplusplus
}
#endif
#endif
