// The absorb / squeeze batch driver as a sequence of wave-uniform PASSES (wide states; pmx_device.hip: sponge_first_kernel,
// permute_listed_kernel, sponge_walk).
//
// Reference semantics (file:line in /root/reference):
//   absorb   src/poseidon/mod.rs:232-254 (mode handling) + absorb_internal :121-150
//   squeeze  src/poseidon/mod.rs:321-341 (mode handling) + squeeze_internal :153-182, incl. the `!= rate` test of :175
//
// A sponge's call is a strict alternation  [permute] move chunk 0, [permute] move chunk 1, ...  where "move" adds input
// elements into the rate portion of the state (absorb) or copies them out of it (squeeze).  Which permutations happen and
// which elements each chunk holds depend only on the sponge's ORIGINAL mode words and the call's length - so pass p of
// the launch loop can recompute its share from those two words and its own number, with no per-sponge scratch:
//
//   pass p  =  move the chunk in front of the sponge's OWN p-th permutation (as the state is loaded, or in global memory by
//              the owning lane - dynamic addressing is free there)  ->  that permutation
//
// and a sponge's mode words are rewritten only when the call is over for it.  Passes are numbered by each sponge's own permutations, not by
// chunk: a sponge whose mode asks for a permutation up front (Squeezing, or an index equal to the rate) and one that
// absorbs first both run their p-th permutation in pass p, so a call costs max-over-sponges permutation launches (2 for
// absorb(11) at rate 8 whatever the modes), not one per chunk boundary any sponge happens to have (3).
// The permutation then runs wave-uniform on the fast engines (the per-lane state machine of absorb_kernel / squeeze_kernel
// left no room for the byte operands of the matrix-core rows) and is kept only by the sponges it is due for.
#pragma once
#include <cstddef>
#include <cstdint>

#include "pmx_field.hpp"

namespace pmx {

struct SpongePass {
    bool permute;        // permutation p of this sponge happens (after the move below)
    uint32_t state_pos;  // move: first state element touched (index into the state, capacity included)
    uint32_t first;      // move: first element of the call's input / output row
    uint32_t count;      // move: elements (0: nothing to move in this pass)
    uint32_t end_index;  // next_absorb_index / next_squeeze_index after the whole call (valid in every pass)
};
// All of it is 32-bit arithmetic (a 64-bit division costs a lane a hundred instructions): a call moves fewer than 2^31
// elements per sponge (kSpongeMaxLen; the launchers refuse longer ones - 64 GiB per sponge).
constexpr size_t kSpongeMaxLen = 0x7fffffffu;

// number of kernel passes (p = 0 .. passes - 1) that cover every sponge of an absorb(len) / squeeze(len) call at this rate:
// the last one only moves the last chunk and rewrites the mode words
// (a sponge permutes at most ceil(len / rate) times in one call - at least once when an absorbing sponge squeezes nothing)
PMX_FN size_t absorb_passes(size_t len, uint32_t rate) { return len == 0 ? 0 : 1 + (len + rate - 1) / rate; }
PMX_FN size_t squeeze_passes(size_t len, uint32_t rate) { return len == 0 ? 2 : 1 + (len + rate - 1) / rate; }

// absorb(len > 0) of a sponge in mode (tag, index): mod.rs:232-254, 121-150
PMX_FN SpongePass absorb_pass(uint32_t tag, uint32_t index, uint32_t len, uint32_t rate, uint32_t capacity, uint32_t pass) {
    if (index > rate) index = rate;                       // device-resident mode words are not validated by the host
    // Squeezing: always permute first, then absorb at 0 (:247-252); Absorbing with a full rate: permute first (:241-246)
    const bool first_perm = tag != PMX_MODE_ABSORBING || index == rate;
    const uint32_t i0 = first_perm ? 0 : index;
    // chunk c holds elements [k(c), k(c+1)): k(0) = 0, k(1) = min(len, rate - i0), k(c+1) = min(len, k(c) + rate)
    auto k = [&](uint32_t c) -> uint32_t {
        if (c == 0) return 0;
        const uint64_t v = (uint64_t)(rate - i0) + (uint64_t)(c - 1) * rate;   // (one 32 x 32 multiply-add)
        return v < len ? (uint32_t)v : len;
    };
    // step q of this sponge = "move chunk q-1, then the permutation in front of chunk q"; a sponge without the up-front
    // permutation has no step 0, so its pass p is step p + 1
    const uint32_t q = pass + (first_perm ? 0 : 1);
    SpongePass r;
    // the rate was filled and more input remains -> permute (:137-148); step 0: the mode's own permutation
    r.permute = q == 0 ? first_perm : k(q) < len;
    r.count = 0;
    r.first = 0;
    r.state_pos = capacity;
    if (q >= 1) {
        const uint32_t lo = k(q - 1), hi = k(q);
        r.first = lo;
        r.count = hi - lo;
        r.state_pos = capacity + (q == 1 ? i0 : 0);
    }
    // the index after the last element: filling the rate exactly does not permute (:126-135), it ends at `rate`
    r.end_index = (i0 + len - 1) % rate + 1;
    return r;
}

// squeeze_native_field_elements(len) of a sponge in mode (tag, index): mod.rs:321-341, 153-182
PMX_FN SpongePass squeeze_pass(uint32_t tag, uint32_t index, uint32_t len, uint32_t rate, uint32_t capacity, uint32_t pass) {
    if (index > rate) index = rate;
    // Absorbing: permute, squeeze from 0 - also for len == 0 (:324-328); Squeezing: permute iff the rate is used up (:330-336)
    const bool first_perm = tag != PMX_MODE_SQUEEZING || index == rate;
    const uint32_t i0 = first_perm ? 0 : index;
    const bool fits = i0 + len <= rate;                    // chunk 0 is the whole output
    const uint32_t take0 = fits ? len : rate - i0;
    // chunk c >= 1 starts at output element o(c) = take0 + (c - 1) rate and exists while o(c) < len
    auto o = [&](uint32_t c) -> uint64_t { return c == 0 ? 0 : (uint64_t)take0 + (uint64_t)(c - 1) * rate; };
    const uint32_t q = pass + (first_perm ? 0 : 1);        // step of this sponge (see absorb_pass)
    SpongePass r;
    if (q == 0) r.permute = first_perm;
    else if (fits || o(q) >= len) r.permute = false;
    else r.permute = q == 1 ? len != rate : true;          // :175 - tested on the remaining length BEFORE it is advanced
    r.count = 0;
    r.first = 0;
    r.state_pos = capacity;
    if (q >= 1) {
        const uint32_t c = q - 1;
        const uint64_t lo = o(c);
        if (c == 0 || (!fits && lo < len)) {
            const uint32_t room = c == 0 ? take0 : rate;
            const uint32_t left = len - (uint32_t)lo;
            r.first = (uint32_t)lo;
            r.count = left < room ? left : room;
            r.state_pos = capacity + (c == 0 ? i0 : 0);
        }
    }
    if (fits) r.end_index = i0 + len;
    else r.end_index = (len - take0 - 1) % rate + 1;
    return r;
}

}  // namespace pmx
