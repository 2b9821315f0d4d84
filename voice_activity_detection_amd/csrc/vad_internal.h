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

// One-time per-device opt-in of the model kernels that need more than 64 KB of LDS; defined in rnn.hip.
int ensure_rnn_attrs();
int ensure_densenet_attrs();   // the same for the stride-1 DenseNet kernels; defined in densenet.hip
int ensure_nicknet_attrs();    // the same for the conv + GRU model's kernels; defined in nicknet.hip

// Looks up, once per device, the address of the built-in histogram of vad_threshold_sweep_noalloc_f32; sweep.hip.
int ensure_sweep_scratch();

// Where the windows of a model launch start.  rows == nullptr: window w starts at x + w * stride floats
// (stride 24 = stride-1 windows of a packed feature matrix, 720 = dense (B,30,24) batch).  Otherwise window w
// starts at feature row rows[w] of a packed (n_rows, 24) matrix (sampled windows); a row index outside
// 0 .. last_start is clamped so that a bad index cannot make the kernel read outside the matrix.
struct WindowMap {
    long stride;
    const int64_t* rows;
    long last_start;
};

__device__ __forceinline__ const float* window_ptr(const float* x, const WindowMap& m, long w) {
    if (m.rows) {
        long r = (long)m.rows[w];
        r = r < 0 ? 0 : (r > m.last_start ? m.last_start : r);
        return x + r * VAD_FEATURES;
    }
    return x + w * m.stride;
}

// Validates the arguments of the *_rows_f32 entry points and builds their map; defined in rnn.hip.
int window_map_for_rows(int64_t n_rows, const int64_t* window_rows_dev, int64_t n_windows, WindowMap* wm);

// 32x32 MFMA accumulator geometry (v_mfma_f32_32x32x2_f32): register i of a
// lane in half hh (= lane >> 5) holds row (i & 3) + 8 * (i >> 2) + 4 * hh of
// the 32-row tile; the column is lane & 31.
static inline __host__ __device__ int mfma32_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

}  // namespace vad
