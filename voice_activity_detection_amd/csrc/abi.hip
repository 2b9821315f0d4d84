// Error channel, init and the synthetic audio generator of libvad_amd.so.

#include <string.h>

#include "vad_internal.h"

namespace vad {

namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x) {      // lowbias32 finaliser
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// Integer-only, stateless: sample n of stream `seed`.  The host twin is
// oracle/synth_ref.py; both must produce identical bits.
__device__ __forceinline__ int synth_sample(uint64_t n, uint32_t seed) {
    const int noise_amp[8] = {0, 0, 48, 48, 640, 640, 2560, 9000};
    const int tone_amp[8] = {0, 0, 0, 1500, 0, 6000, 0, 12000};
    const uint32_t lo = (uint32_t)n, hi = (uint32_t)(n >> 32);
    const uint32_t k1 = hash32(lo ^ hash32(hi + seed * 0x9E3779B9u));
    const uint32_t k2 = hash32(k1 ^ 0x68E31DA4u);
    const int noise = (int)(k1 & 0xFFFF) + (int)(k1 >> 16) + (int)(k2 & 0xFFFF) + (int)(k2 >> 16) - 131070;
    const uint64_t seg = n / 4000u;
    const uint32_t hs = hash32((uint32_t)seg ^ hash32((uint32_t)(seg >> 32) ^ (seed + 0x51ED270Bu)));
    const int level = (int)(hs & 7u);
    const uint32_t fstep = (((hs >> 8) & 0x3FFu) + 16u) << 20;
    const uint32_t phase = lo * fstep;
    const int p = (int)(phase >> 16);
    const int tri = 2 * (p < 32768 ? p : 65535 - p) - 32767;
    const long v = (((long)noise * noise_amp[level]) >> 16) + (((long)tri * tone_amp[level]) >> 16);
    return (int)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

__global__ __launch_bounds__(256) void synth_kernel(int16_t* __restrict__ out, long first, long n, uint32_t seed) {
    // two samples per thread so that stores are 4 bytes wide when `out` allows it
    const long pair = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = 2 * pair; i < n; i += 2 * stride) {
        out[i] = (int16_t)synth_sample((uint64_t)(first + i), seed);
        if (i + 1 < n) out[i + 1] = (int16_t)synth_sample((uint64_t)(first + i + 1), seed);
    }
}

}  // namespace
}  // namespace vad

extern "C" int vad_abi_version(void) { return 1; }

extern "C" const char* vad_last_error(void) { return vad::g_err; }

extern "C" int vad_init(void) {
    int rc = vad::ensure_mfcc_tables();
    if (rc != VAD_OK) return rc;
    rc = vad::ensure_rnn_attrs();
    if (rc != VAD_OK) return rc;
    rc = vad::ensure_densenet_attrs();
    if (rc != VAD_OK) return rc;
    rc = vad::ensure_nicknet_attrs();
    if (rc != VAD_OK) return rc;
    return vad::ensure_sweep_scratch();
}

extern "C" int vad_synth_audio_i16(int16_t* out_dev, int64_t first_sample, int64_t n_samples, uint32_t seed,
                                   void* stream) {
    using namespace vad;
    VAD_REQUIRE(out_dev, "null pointer argument");
    VAD_REQUIRE(first_sample >= 0 && n_samples >= 0, "negative sample range");
    if (n_samples == 0) return VAD_OK;
    long blocks = (n_samples / 2 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out_dev,
                       (long)first_sample, (long)n_samples, seed);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}
