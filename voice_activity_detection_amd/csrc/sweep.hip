// Threshold sweep over the posteriors: for every threshold t_k of an ascending table, how many
// scores of each class satisfy score >= t_k.  One pass over the scores gives the confusion matrix of
// every threshold at once, which is what the reference's FAR-at-fixed-FRR search (fix_frr,
// core/train.py:489-528: apply_threshold `y >= t` + confusion_matrix per candidate t, t *= 1.1)
// recomputes from scratch for each of its ~100-200 candidates.
//
// The comparison is exact, in double precision (a float32 score against a float64 threshold).  The
// reference's `y >= t` (numpy.float32 against a Python float) is a float32 comparison under numpy >= 2:
// callers that want that rule round their thresholds to float32 first.
// Integer counting only: results are exact and independent of the launch geometry.

#include <mutex>

#include "vad_internal.h"

namespace vad {
namespace {

constexpr int kMaxThresholds = VAD_MAX_THRESHOLDS;                    // 16 KB of thresholds + 16 KB of counters in LDS
constexpr int kSweepThreads = 512;

// bin = number of thresholds <= score (0 .. T), searched inside [lo, hi]
__device__ __forceinline__ int bin_search(double score, const double* thr, int lo, int hi) {
    while (lo < hi) {                                       // invariant: thr[lo-1] <= score < thr[hi]
        const int mid = (lo + hi) >> 1;
        if (thr[mid] <= score) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// A non-negative finite float32 with exponent field e lies in [edge(e), edge(e+1)), edge(e) = 2^(e-127)
// (edge(0) = 0), so its bin lies in [lut[e], lut[e+1]] with lut[e] = #thresholds <= edge(e): one table
// read narrows the search to the few thresholds of the score's binade.  Negative scores, infinities and
// NaN (which compares false: bin 0) take the full search.
__device__ __forceinline__ int bin_of(float score, const double* thr, const unsigned short* lut, int T) {
    const unsigned bits = __float_as_uint(score);
    const unsigned e = bits >> 23;                          // sign bit included: e > 254 for negatives / inf / NaN
    if (e <= 254u) return bin_search((double)score, thr, lut[e], lut[e + 1]);
    return bin_search((double)score, thr, 0, T);
}

// One LDS increment per distinct key of the wave for its two most common keys (posteriors cluster near
// 0 and 1, which would otherwise serialise 64 same-address LDS atomics), plain LDS atomics for the rest.
__device__ __forceinline__ void wave_count(unsigned* hist, int key, bool valid) {
    unsigned long long todo = __ballot(valid);
#pragma unroll 1
    for (int round = 0; round < 2 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lkey = __shfl(key, leader);
        const unsigned long long same = __ballot(valid && key == lkey) & todo;
        if ((threadIdx.x & 63) == leader) atomicAdd(&hist[lkey], (unsigned)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> (threadIdx.x & 63)) & 1) atomicAdd(&hist[key], 1u);
}

// hist_out[(T+1)][2]: scores of class c whose bin is b.  VEC: score_stride == 1 and both arrays aligned
// for 16-byte / 4-byte loads of four elements per lane.
template <bool VEC>
__global__ __launch_bounds__(kSweepThreads) void sweep_hist_kernel(const float* __restrict__ score, long score_stride,
                                                                   const uint8_t* __restrict__ label, long n,
                                                                   const double* __restrict__ thresholds, int T,
                                                                   unsigned long long* __restrict__ hist_out) {
    __shared__ double thr[kMaxThresholds];
    __shared__ unsigned hist[2 * (kMaxThresholds + 1)];
    __shared__ unsigned short lut[256];
    for (int i = threadIdx.x; i < T; i += kSweepThreads) thr[i] = thresholds[i];
    for (int i = threadIdx.x; i < 2 * (T + 1); i += kSweepThreads) hist[i] = 0;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int e = threadIdx.x;                          // edge(255) = 2^128 stands for +inf's binade start
        const double edge = e == 0 ? 0.0 : ldexp(1.0, e - 127);
        lut[e] = (unsigned short)bin_search(edge, thr, 0, T);
    }
    __syncthreads();
    // a block's chunk is bounded so that its 32-bit LDS counters cannot overflow; chunks are multiples of 4
    const long per_block = (((n + gridDim.x - 1) / gridDim.x) + 3) & ~3L;
    const long lo = (long)blockIdx.x * per_block;
    long hi = lo + per_block;
    if (hi > n) hi = n;
    long base = lo;
    if (VEC) {
        for (; base + 4 * kSweepThreads <= hi; base += 4 * kSweepThreads) {
            const long i = base + 4 * threadIdx.x;
            const float4 sv = *reinterpret_cast<const float4*>(score + i);
            const uchar4 lv = *reinterpret_cast<const uchar4*>(label + i);
            wave_count(hist, 2 * bin_of(sv.x, thr, lut, T) + (lv.x ? 1 : 0), true);
            wave_count(hist, 2 * bin_of(sv.y, thr, lut, T) + (lv.y ? 1 : 0), true);
            wave_count(hist, 2 * bin_of(sv.z, thr, lut, T) + (lv.z ? 1 : 0), true);
            wave_count(hist, 2 * bin_of(sv.w, thr, lut, T) + (lv.w ? 1 : 0), true);
        }
    }
    for (; base < hi; base += kSweepThreads) {              // uniform trip count inside a wave
        const long i = base + threadIdx.x;
        const bool valid = i < hi;
        int key = 0;
        if (valid) key = 2 * bin_of(score[i * score_stride], thr, lut, T) + (label[i] ? 1 : 0);
        wave_count(hist, key, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (T + 1); i += kSweepThreads)
        if (hist[i]) atomicAdd(&hist_out[i], (unsigned long long)hist[i]);
}

// ge[k][c] = sum over bins b > k of hist[b][c]  (scores >= thresholds[k]); total[c] = all bins.  One block.
__global__ __launch_bounds__(256) void sweep_suffix_kernel(const unsigned long long* __restrict__ hist, int T,
                                                           long long* __restrict__ ge, long long* __restrict__ total) {
    __shared__ long long part[256][2];
    const int per = (T + 1 + 255) / 256;                    // bins per thread, contiguous
    const int b0 = threadIdx.x * per;
    long long s0 = 0, s1 = 0;
    for (int b = b0; b < b0 + per && b <= T; ++b) {
        s0 += (long long)hist[2 * b];
        s1 += (long long)hist[2 * b + 1];
    }
    part[threadIdx.x][0] = s0;
    part[threadIdx.x][1] = s1;
    __syncthreads();
    long long a0 = 0, a1 = 0;                                // everything in later threads' bins
    for (int t = threadIdx.x + 1; t < 256; ++t) {
        a0 += part[t][0];
        a1 += part[t][1];
    }
    for (int b = (b0 + per - 1 <= T ? b0 + per - 1 : T); b >= b0; --b) {
        // a0/a1 = counts in bins > b
        if (b < T) {
            ge[2 * b] = a0;
            ge[2 * b + 1] = a1;
        }
        a0 += (long long)hist[2 * b];
        a1 += (long long)hist[2 * b + 1];
    }
    if (threadIdx.x == 0) {
        total[0] = a0;
        total[1] = a1;
    }
}

}  // namespace
}  // namespace vad

namespace vad {
namespace {

// The shared launch sequence: clear the (T+1) x 2 histogram, count, suffix-sum.  `hist` is device memory of
// at least 2 * (T + 1) 64-bit words: the caller's workspace or this library's own scratch array.
int sweep_launch(const float* score_dev, int64_t score_stride, const uint8_t* label_dev, int64_t n,
                 const double* thresholds_dev, int n_thresholds, unsigned long long* hist, int64_t* ge_dev,
                 int64_t* total_dev, hipStream_t st) {
    VAD_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * 2 * (n_thresholds + 1), st));
    if (n > 0) {
        long blocks = (n + kSweepThreads * 64 - 1) / (kSweepThreads * 64);
        if (blocks > 1024) blocks = 1024;
        // 32-bit per-block counters: keep a block's chunk below 2^32 elements
        const long min_blocks = (long)((n >> 31) + 1);
        if (blocks < min_blocks) blocks = min_blocks;
        const bool vec = score_stride == 1 && (reinterpret_cast<uintptr_t>(score_dev) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(label_dev) & 3) == 0;
        if (vec)
            hipLaunchKernelGGL(sweep_hist_kernel<true>, dim3((unsigned)blocks), dim3(kSweepThreads), 0, st, score_dev,
                               (long)score_stride, label_dev, (long)n, thresholds_dev, n_thresholds, hist);
        else
            hipLaunchKernelGGL(sweep_hist_kernel<false>, dim3((unsigned)blocks), dim3(kSweepThreads), 0, st, score_dev,
                               (long)score_stride, label_dev, (long)n, thresholds_dev, n_thresholds, hist);
        VAD_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(sweep_suffix_kernel, dim3(1), dim3(256), 0, st, hist, n_thresholds,
                       reinterpret_cast<long long*>(ge_dev), reinterpret_cast<long long*>(total_dev));
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

int check_sweep_args(const float* score_dev, int64_t score_stride, const uint8_t* label_dev, int64_t n,
                     const double* thresholds_dev, int n_thresholds, const int64_t* ge_dev, const int64_t* total_dev) {
    VAD_REQUIRE(n >= 0 && score_stride >= 1, "n must be >= 0 and score_stride >= 1");
    VAD_REQUIRE(n_thresholds >= 1 && n_thresholds <= kMaxThresholds, "n_thresholds must be in 1..%d, got %d",
                kMaxThresholds, n_thresholds);
    VAD_REQUIRE(thresholds_dev && ge_dev && total_dev && (n == 0 || (score_dev && label_dev)), "null pointer argument");
    return VAD_OK;
}

}  // namespace
}  // namespace vad

// The histogram of the entry point that takes no workspace: one array per GPU, inside the code object, so no
// call ever allocates it.  In host code `scratch` is only a handle for hipGetSymbolAddress; kernels and memsets
// get the device address looked up below.
static __device__ __align__(256) unsigned long long scratch[2 * (VAD_MAX_THRESHOLDS + 1)];

namespace vad {
namespace {
constexpr int kMaxSweepDevices = 64;
std::mutex g_scratch_mu;
unsigned long long* g_scratch_ptr[kMaxSweepDevices] = {nullptr};      // per device: that GPU's copy of `scratch`
}  // namespace

// Device address of `scratch` on the current device, looked up once per device.
int sweep_scratch(unsigned long long** out) {
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_REQUIRE(dev >= 0 && dev < kMaxSweepDevices, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        VAD_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        VAD_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address for the sweep scratch");
        g_scratch_ptr[dev] = static_cast<unsigned long long*>(p);
    }
    if (out) *out = g_scratch_ptr[dev];
    return VAD_OK;
}

int ensure_sweep_scratch() { return sweep_scratch(nullptr); }

}  // namespace vad

extern "C" int vad_threshold_sweep_f32(const float* score_dev, int64_t score_stride, const uint8_t* label_dev,
                                       int64_t n, const double* thresholds_dev, int n_thresholds,
                                       unsigned long long* hist_dev, int64_t* ge_dev, int64_t* total_dev, void* stream) {
    using namespace vad;
    const int rc = check_sweep_args(score_dev, score_stride, label_dev, n, thresholds_dev, n_thresholds, ge_dev, total_dev);
    if (rc != VAD_OK) return rc;
    VAD_REQUIRE(hist_dev, "null pointer argument");
    return sweep_launch(score_dev, score_stride, label_dev, n, thresholds_dev, n_thresholds, hist_dev, ge_dev, total_dev,
                        reinterpret_cast<hipStream_t>(stream));
}

extern "C" int vad_threshold_sweep_noalloc_f32(const float* score_dev, int64_t score_stride, const uint8_t* label_dev,
                                               int64_t n, const double* thresholds_dev, int n_thresholds,
                                               int64_t* ge_dev, int64_t* total_dev, void* stream) {
    using namespace vad;
    const int rc = check_sweep_args(score_dev, score_stride, label_dev, n, thresholds_dev, n_thresholds, ge_dev, total_dev);
    if (rc != VAD_OK) return rc;
    unsigned long long* hist = nullptr;
    const int rs = sweep_scratch(&hist);                   // a table lookup after the device's first call / vad_init
    if (rs != VAD_OK) return rs;
    return sweep_launch(score_dev, score_stride, label_dev, n, thresholds_dev, n_thresholds, hist, ge_dev, total_dev,
                        reinterpret_cast<hipStream_t>(stream));
}
