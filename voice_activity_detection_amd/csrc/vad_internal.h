// Internal helpers shared by the translation units of libvad_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "vad_amd.h"

namespace vad {

// Thread-local message returned by vad_last_error().
void set_error(const char* fmt, ...);

#define VAD_HIP_CHECK(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            vad::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                           __FILE__, __LINE__);                                    \
            return VAD_EHIP;                                                       \
        }                                                                          \
    } while (0)

#define VAD_REQUIRE(cond, ...)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            vad::set_error(__VA_ARGS__);                                           \
            return VAD_EINVAL;                                                     \
        }                                                                          \
    } while (0)

// Per-device constant tables of the MFCC front end; defined in mfcc.hip.
int ensure_mfcc_tables();

// delta rows [g_first, g_first + g_count) from the MFCC rows of a sliced stream; defined in mfcc.hip.
int launch_delta(const float* mfcc_dev, long mfcc_ld, long n_frames, long frames_per_slice, float* delta_dev,
                 long delta_ld, long g_first, long g_count, hipStream_t st);

// One-time per-device opt-in of the model kernels that need more than 64 KB of LDS; defined in rnn.hip.
int ensure_rnn_attrs();

// 32x32 MFMA accumulator geometry (v_mfma_f32_32x32x2_f32): register i of a
// lane in half hh (= lane >> 5) holds row (i & 3) + 8 * (i >> 2) + 4 * hh of
// the 32-row tile; the column is lane & 31.
static inline __host__ __device__ int mfma32_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

}  // namespace vad
