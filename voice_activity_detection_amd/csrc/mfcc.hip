// MFCC + delta front end for gfx950.
//
// One 64-lane wave computes one 30 ms frame: it loads the 1920-sample analysis
// window (4 frames of 480 int16 samples) together with the samples one frame
// earlier, applies the lag-480 pre-emphasis, runs a 2048-point real FFT as a
// 1024-point complex FFT held 16 points per lane, converts to a power
// spectrum, reduces it over the 27 mel segments, takes the log and applies the
// lifted DCT-II.  The arithmetic follows python_speech_features.mfcc as called
// at core/process_data.py:272-276 of the reference (rectangular window,
// nfft 2048, 26 filters, ortho DCT, lifter 22, c0 dropped).
//
// FFT decomposition (N = 1024 = 4 x 16 x 16), all index algebra mirrored by
// tools/fft_lane_model.py:
//   pass A  lane L holds z[L + 64 jj]; radix-4 over jj >> 2 in registers; twiddle W1024^(n2 k1)
//   xchg 1  through LDS: lane (k1, c) gathers the 16 points n2 = c + 16 m of sub-transform k1
//   pass B  radix-16 over m in registers; twiddle W256^(c q1)
//   xchg 2  through LDS: lane (k1, q1) gathers the 16 points c
//   pass C  radix-16 over c in registers
//   result  lane 16 k1 + q1, register q2 holds Z[k1 + 4 q1 + 64 q2]
// The real-input untangle works on the pairs (k, 1024 - k), k < 512, that share Z[k] and Z[1024 - k]:
// the partner value is a register of another lane (ds_bpermute, no LDS image of Z); the power spectrum
// is left in LDS for the mel stage.
//
// Occupancy is the design constraint: a wave stalls on its own LDS exchanges, so the kernel keeps three
// waves per SIMD (<= 168 VGPRs, 24 KB of LDS per 2-wave block): the input is read straight from global
// memory (no block staging buffer), one of the three twiddle sets lives in a block-shared LDS table.
//
// Precision: FFT, power and mel sums in fp32; filter energies, log and DCT in
// fp64 (26 logs and 312 fused multiply-adds per frame), because an fp32 log of
// an energy near e^25 already carries 2e-6 absolute error before the DCT.

#include <math.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "mfcc_device.h"

namespace vad {

using namespace fe;

namespace {

constexpr int kWavesPerBlock = 2;                            // small blocks schedule best (measured 2 < 3 < 4 << 6 waves)
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;
constexpr int kTwBytes = kTwFloat2s * 8;                     // block-shared twiddle tables
constexpr int kSmemBytes = kWavesPerBlock * kWaveLdsBytes + kTwBytes;
constexpr int kWavesPerSimd = 3;                             // register budget 168; a block needs 24 KB of LDS

// Where frame g of the launch lives: slice s, frame j inside it.
struct FramePos {
    long s;
    long j;
};

// grid: one block per 8 consecutive frames of the launch (slices are not padded to blocks: a block may
// straddle slices, each frame finds its own slice).  Wave w takes frames w, w+2, w+4, w+6 of the block, so
// the waves walk neighbouring rows at the same time and the 4x re-read of every input row (hop 480,
// window 1920, plus the pre-emphasis lag) is served by the caches.
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void mfcc_kernel(const int16_t* __restrict__ in,
                                                                 const MfccTables* __restrict__ tab, long n_frames,
                                                                 long frames_per_slice, long slice_stride,
                                                                 float* __restrict__ out, long out_ld) {
    __shared__ __align__(16) unsigned char smem[kSmemBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    unsigned char* wbase = smem + wave * kWaveLdsBytes;
    float2* ex = reinterpret_cast<float2*>(wbase);                          // exchange 1 / 2, power spectrum
    float* en = reinterpret_cast<float*>(wbase + kExBytes);                  // [4 frames][32]

    // block-uniform: slice and in-slice offset of the block's first frame
    const long g0 = (long)blockIdx.x * kFramesPerBlock;
    const long s0 = g0 / frames_per_slice;
    const long r0 = g0 - s0 * frames_per_slice;
    auto locate = [&](int jl) {                             // frame g0 + jl (wave-uniform)
        FramePos p;
        const long off = r0 + jl;
        if (off < frames_per_slice) {
            p.s = s0;
            p.j = off;
        } else if (frames_per_slice >= kFramesPerBlock) {   // at most one slice boundary inside a block
            p.s = s0 + 1;
            p.j = off - frames_per_slice;
        } else {
            const long ds = off / frames_per_slice;
            p.s = s0 + ds;
            p.j = off - ds * frames_per_slice;
        }
        return p;
    };
    // aligned row r of a slice = its frame r - 3; frame j reads rows j .. j+3 and their predecessors
    // j-1 .. j+2 (row -1 does not exist: zero_first_pred).  Every slice holds its frames + 3 rows.
    auto lane_ptr = [&](const FramePos& p) {
        return reinterpret_cast<const int*>(in + p.s * slice_stride) + p.j * (kFrame / 2) + lane;
    };

    LaneConst K;
    load_lane_const(K, tab, lane);
    const float2* tws = reinterpret_cast<const float2*>(smem + kWavesPerBlock * kWaveLdsBytes);

    RawFrame raw;
    {
        // tw1 is the first member of MfccTables
        const float4* src = reinterpret_cast<const float4*>(tab);
        float4* dst = reinterpret_cast<float4*>(smem + kWavesPerBlock * kWaveLdsBytes);
        for (int i = threadIdx.x; i < kTwFloat2s / 2; i += kThreads) dst[i] = src[i];
        __syncthreads();
    }

    int nfr = 0;                                            // frames this wave produced (wave-uniform)
#pragma unroll 1
    for (int it = 0; it < kFramesPerWave; ++it) {
        const int jl = wave + it * kWavesPerBlock;
        if (g0 + jl >= n_frames) break;
        nfr = it + 1;
        const FramePos p = locate(jl);
        // (fetching the wave's next frame under the mel stage instead was measured slower: it costs
        // 30 live registers and the third wave per SIMD already covers the load latency)
        frame_load(raw, lane_ptr(p), p.j == 0, lane);
        frame_energies(raw, tws, K, ex, en + it * 32, lane);
    }
    cepstra_tail(en, reinterpret_cast<double*>(wbase), &tab->dct[0][0], nfr, lane, out, out_ld, g0 + wave,
                 kWavesPerBlock);
}

// delta[g] = (-2 f[g-2] - f[g-1] + f[g+1] + 2 f[g+2]) / 10 with the slice's own edge rows
// replicated (python_speech_features.delta(feat, 2), core/process_data.py:277).
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ mfcc, long mfcc_ld, long n_frames,
                                                    long frames_per_slice, float* __restrict__ delta, long delta_ld,
                                                    long g_first, long g_count) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g_count * kNCep) return;
    const long g = g_first + idx / kNCep;
    const int c = (int)(idx % kNCep);
    const long s = g / frames_per_slice;
    const long lo = s * frames_per_slice;
    long hi = lo + frames_per_slice - 1;
    if (hi > n_frames - 1) hi = n_frames - 1;
    auto at = [&](long r) {
        r = r < lo ? lo : (r > hi ? hi : r);
        return mfcc[r * mfcc_ld + c];
    };
    const float d = (2.0f * (at(g + 2) - at(g - 2)) + (at(g + 1) - at(g - 1))) / 10.0f;
    delta[g * delta_ld + c] = d;
}

// ---------------------------------------------------------------------------------------------
// host side: tables

double hz2mel(double hz) { return 2595.0 * log10(1.0 + hz / 700.0); }
double mel2hz(double mel) { return 700.0 * (pow(10.0, mel / 2595.0) - 1.0); }

void build_tables(MfccTables& t) {
    memset(&t, 0, sizeof(t));
    const double pi = 3.14159265358979323846;
    for (int k1 = 1; k1 < 4; ++k1)
        for (int b = 0; b < 4; ++b)
            for (int l = 0; l < 64; ++l) {
                const double a = -2.0 * pi * (double)((l + 64 * b) * k1) / 1024.0;
                t.tw1[(k1 - 1) * 4 + b][l] = make_float2((float)cos(a), (float)sin(a));
            }
    for (int q1 = 1; q1 < 16; ++q1)
        for (int c = 0; c < 16; ++c) {
            const double a = -2.0 * pi * (double)(c * q1) / 256.0;
            t.tw2[q1 - 1][c] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int j = 0; j < 8; ++j)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * pi * (double)((l >> 4) + 4 * (l & 15) + 64 * j) / 2048.0;
            t.twr[j][l] = make_float2((float)cos(a), (float)sin(a));
        }

    // 28 bin edges of the 26 triangular filters: floor((nfft + 1) * hz / fs) on a mel-linear grid.
    int edge[kNFilt + 2];
    const double lowmel = hz2mel(0.0), highmel = hz2mel(VAD_SAMPLE_RATE / 2.0);
    for (int i = 0; i < kNFilt + 2; ++i) {
        const double mel = lowmel + (highmel - lowmel) * (double)i / (double)(kNFilt + 1);
        edge[i] = (int)floor((2048 + 1) * mel2hz(mel) / VAD_SAMPLE_RATE);
    }
    // Segment s = bins [edge[s], edge[s+1]) feeds the rising edge of filter s and the falling
    // edge of filter s-1.  Lanes are dealt to segments so that no lane sums more than kMaxRun bins.
    int seg_first[kNFilt + 1], seg_n[kNFilt + 1];
    int lane = 0;
    for (int s = 0; s <= kNFilt; ++s) {
        const int len = edge[s + 1] - edge[s];
        const int nl = (len + kMaxRun - 1) / kMaxRun;
        const int chunk = (len + nl - 1) / nl;
        seg_first[s] = lane;
        seg_n[s] = nl;
        for (int i = 0; i < nl; ++i, ++lane) {
            if (lane >= 64) abort();
            const int st = edge[s] + i * chunk;
            int cnt = edge[s + 1] - st;
            if (cnt > chunk) cnt = chunk;
            if (cnt < 0) cnt = 0;
            t.run_start[lane] = st;
            t.run_count[lane] = cnt;
            t.run_b0[lane] = edge[s];
            t.run_b1[lane] = edge[s + 1];
            t.run_inv[lane] = (float)(1.0 / (double)len);
        }
    }
    for (int f = 0; f < kNFilt; ++f) {
        t.rise_first[f] = seg_first[f];
        t.rise_n[f] = seg_n[f];
        t.fall_first[f] = seg_first[f + 1];
        t.fall_n[f] = seg_n[f + 1];
    }
    for (int n = 0; n < kNFilt; ++n)
        for (int k = 1; k <= kNCep; ++k) {
            const double lift = 1.0 + 11.0 * sin(pi * (double)k / 22.0);
            t.dct[n][k - 1] = sqrt(2.0 / kNFilt) * lift * cos(pi * (double)k * (2.0 * n + 1.0) / (2.0 * kNFilt));
        }
}

constexpr int kMaxDevices = 64;
std::mutex g_mu;
MfccTables* g_tables[kMaxDevices] = {nullptr};

}  // namespace

const fe::MfccTables* mfcc_tables_for_current_device() {
    if (ensure_mfcc_tables() != VAD_OK) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return g_tables[dev];
}

int ensure_mfcc_tables() {
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_tables[dev]) return VAD_OK;
    static MfccTables host;            // built once, copied per device
    static bool built = false;
    if (!built) {
        build_tables(host);
        built = true;
    }
    MfccTables* d = nullptr;
    VAD_HIP_CHECK(hipMalloc(&d, sizeof(MfccTables)));
    VAD_HIP_CHECK(hipMemcpy(d, &host, sizeof(MfccTables), hipMemcpyHostToDevice));
    g_tables[dev] = d;
    return VAD_OK;
}

}  // namespace vad

namespace vad {
// delta rows [g_first, g_first + g_count) of a stream of n_frames frames cut into frames_per_slice slices
int launch_delta(const float* mfcc_dev, long mfcc_ld, long n_frames, long frames_per_slice, float* delta_dev,
                 long delta_ld, long g_first, long g_count, hipStream_t st) {
    if (g_count <= 0) return VAD_OK;
    const long dblocks = (g_count * kNCep + 255) / 256;
    VAD_REQUIRE(dblocks <= 0x7FFFFFFFL, "too many frames for one launch");
    hipLaunchKernelGGL(delta_kernel, dim3((unsigned)dblocks), dim3(256), 0, st, mfcc_dev, mfcc_ld, n_frames,
                       frames_per_slice, delta_dev, delta_ld, g_first, g_count);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}
}  // namespace vad

extern "C" int vad_mfcc_delta_f32(const int16_t* frames_dev, int64_t n_slices, int64_t frames_per_slice,
                                  int64_t last_slice_frames, int64_t slice_stride, float* mfcc_dev, int64_t mfcc_ld,
                                  float* delta_dev, int64_t delta_ld, void* stream) {
    using namespace vad;
    VAD_REQUIRE(frames_dev && mfcc_dev && delta_dev, "null pointer argument");
    VAD_REQUIRE(n_slices >= 1 && frames_per_slice >= 1, "n_slices and frames_per_slice must be >= 1");
    VAD_REQUIRE(last_slice_frames >= 1 && last_slice_frames <= frames_per_slice,
                "last_slice_frames must be in 1..frames_per_slice");
    VAD_REQUIRE(mfcc_ld >= kNCep && delta_ld >= kNCep, "leading dimensions must be >= 12");
    VAD_REQUIRE(slice_stride >= 0 && (slice_stride % 2) == 0, "slice_stride must be even and non-negative");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(frames_dev) & 3) == 0, "frames_dev must be 4-byte aligned");
    VAD_REQUIRE(n_slices < (int64_t(1) << 40) && frames_per_slice < (int64_t(1) << 40), "size out of range");
    int rc = ensure_mfcc_tables();
    if (rc != VAD_OK) return rc;
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    const long n_frames = (long)((n_slices - 1) * frames_per_slice + last_slice_frames);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long blocks = (n_frames + kFramesPerBlock - 1) / kFramesPerBlock;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many frames for one launch");
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, frames_dev, g_tables[dev], n_frames,
                       (long)frames_per_slice, (long)slice_stride, mfcc_dev, (long)mfcc_ld);
    VAD_HIP_CHECK(hipGetLastError());
    return launch_delta(mfcc_dev, mfcc_ld, n_frames, frames_per_slice, delta_dev, delta_ld, 0, n_frames, st);
}
