// int16 noise/speech mix ahead of the front end: the arithmetic of
// noise_track.overlay(speech_track, gain_during_overlay=dB) at core/process_data.py:293 of the
// reference (pydub -> audioop.mul on the overlaid part of the noise, then audioop.add):
//     g[i]   = clamp16(floor((double)noise[i] * factor))        when a gain is given
//     out[i] = clamp16(g[i] + speech[i])                        for i < n_speech
//     out[i] = noise[i]                                         for the rest of the noise track
// HBM-bound: 6 bytes per mixed sample; 16-byte accesses when the three pointers allow it.

#include "vad_internal.h"

namespace vad {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // 8 samples; a type the nontemporal builtins take

__device__ __forceinline__ int clamp16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// audioop's bound-then-floor of a scaled sample
__device__ __forceinline__ int scale_sample(int x, double factor) {
    double v = (double)x * factor;
    if (v > 32767.0) v = 32767.0;
    else if (v < -32767.0) v = -32768.0;
    return (int)floor(v);
}

__device__ __forceinline__ int mix(int noise, int speech, bool gain, double factor) {
    return clamp16((gain ? scale_sample(noise, factor) : noise) + speech);
}

__device__ __forceinline__ unsigned mix_pair(unsigned n, unsigned s, bool gain, double factor) {
    const int lo = mix((int)(short)(n & 0xFFFF), (int)(short)(s & 0xFFFF), gain, factor);
    const int hi = mix((int)n >> 16, (int)s >> 16, gain, factor);
    return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
}

// n_vec groups of 8 samples, all inside the overlaid part
__global__ __launch_bounds__(256) void overlay_vec_kernel(const u32x4* __restrict__ noise, const u32x4* __restrict__ speech,
                                                          long n_vec, int gain, double factor, u32x4* __restrict__ out) {
    const long stride = (long)gridDim.x * blockDim.x;
    constexpr int U = 4;                                    // independent 16-byte loads in flight per lane and input
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n_vec; i += U * stride) {
        u32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = __builtin_nontemporal_load(noise + i + u * stride);
            b[u] = __builtin_nontemporal_load(speech + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 r;
            r.x = mix_pair(a[u].x, b[u].x, gain, factor);
            r.y = mix_pair(a[u].y, b[u].y, gain, factor);
            r.z = mix_pair(a[u].z, b[u].z, gain, factor);
            r.w = mix_pair(a[u].w, b[u].w, gain, factor);
            __builtin_nontemporal_store(r, out + i + u * stride);
        }
    }
    for (; i < n_vec; i += stride) {
        const u32x4 a = noise[i], b = speech[i];
        u32x4 r;
        r.x = mix_pair(a.x, b.x, gain, factor);
        r.y = mix_pair(a.y, b.y, gain, factor);
        r.z = mix_pair(a.z, b.z, gain, factor);
        r.w = mix_pair(a.w, b.w, gain, factor);
        out[i] = r;
    }
}

// samples [first, n_total): mixed below n_mix, copied above
__global__ __launch_bounds__(256) void overlay_scalar_kernel(const int16_t* __restrict__ noise,
                                                             const int16_t* __restrict__ speech, long first, long n_mix,
                                                             long n_total, int gain, double factor,
                                                             int16_t* __restrict__ out) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = first + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += stride)
        out[i] = i < n_mix ? (int16_t)mix(noise[i], speech[i], gain, factor) : noise[i];
}

}  // namespace
}  // namespace vad

extern "C" int vad_overlay_i16(const int16_t* noise_dev, int64_t n_noise, const int16_t* speech_dev, int64_t n_speech,
                               int apply_gain, double gain_factor, int16_t* out_dev, void* stream) {
    using namespace vad;
    VAD_REQUIRE(n_noise >= 0 && n_speech >= 0, "negative sample count");
    if (n_noise == 0) return VAD_OK;
    VAD_REQUIRE(noise_dev && out_dev && (speech_dev || n_speech == 0), "null pointer argument");
    VAD_REQUIRE(!apply_gain || (gain_factor == gain_factor && gain_factor >= 0.0), "gain factor must be a number >= 0");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long n_mix = (long)(n_speech < n_noise ? n_speech : n_noise);      // the overlay ends with the noise track
    const bool aligned = ((reinterpret_cast<uintptr_t>(noise_dev) | reinterpret_cast<uintptr_t>(speech_dev) |
                           reinterpret_cast<uintptr_t>(out_dev)) & 15) == 0;
    const long n_vec = aligned ? n_mix / 8 : 0;
    if (n_vec > 0) {
        long blocks = (n_vec + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(overlay_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const u32x4*>(noise_dev), reinterpret_cast<const u32x4*>(speech_dev), n_vec,
                           apply_gain, gain_factor, reinterpret_cast<u32x4*>(out_dev));
        VAD_HIP_CHECK(hipGetLastError());
    }
    const long rest = (long)n_noise - n_vec * 8;
    if (rest > 0) {
        long blocks = (rest + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(overlay_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, noise_dev, speech_dev,
                           n_vec * 8, n_mix, (long)n_noise, apply_gain, gain_factor, out_dev);
        VAD_HIP_CHECK(hipGetLastError());
    }
    return VAD_OK;
}
