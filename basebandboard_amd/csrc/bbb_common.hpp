// bbb_common.hpp -- shared host-side plumbing for libbbb_hip.so (error codes, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/bbb.h"

#include "custom_abi.hpp"

namespace bbb {

// per-thread text of the last failure (returned by bbb_last_error_detail)
std::string &last_error();

inline int fail(int code, const std::string &what) {
    last_error() = what;
    return code;
}

#define BBB_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ::bbb::fail(BBB_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Select `device` after checking that it exists and is a gfx950 part: there is no other
// execution path in this library.
int use_device(int device);

constexpr int kWave = 64;   // CDNA wavefront

// Timing-experiment knobs (kernel variants selected through the environment) exist only in builds made with
// -DBBB_EXPERIMENTS; the shipped library ignores the environment: no variable can change what it computes.
#ifdef BBB_EXPERIMENTS
inline int env_knob(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}
#else
constexpr int env_knob(const char *, int dflt) { return dflt; }
#endif

// ---- PRBS entry points implemented in prbs_kernels.hip -------------------------------------
int prbs_fill_launch(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                     uint64_t *dst, hipStream_t st, int nt_stores = 0);
int prbs_check_launch(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                      const uint64_t *src, uint64_t *nerr_dev, hipStream_t st);
int prbs_detector_launch(int k, const uint8_t *bits, uint64_t nstreams, uint64_t n, uint8_t *err,
                         uint8_t *reload, hipStream_t st);

// detector_kernels.hip
int prbs_detector_stream_launch(int k, const uint64_t *src, uint64_t nbits, uint64_t *err, uint64_t *reload,
                                bbb_detector_stats *stats, uint64_t chunk_bits, uint64_t warm_bits, hipStream_t st);

// search_kernels.hip
int lutopt_search_launch(int k, uint64_t seed, uint64_t first, uint64_t count, uint64_t *found, uint16_t *taps_out,
                         uint32_t *row_off_out, bbb_search_stats *stats, hipStream_t st);

}  // namespace bbb
