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
// FFT decomposition (N = 1024 = 16 x 16 x 4), all index algebra mirrored by
// tools/fft_lane_model.py:
//   pass A  lane L holds z[L + 64 jj]; radix-16 over jj in registers; twiddle W1024^(L k1)
//   xchg    through LDS: lane (k1, a) = 4 k1 + a gathers the 16 points n2 = a + 4 b of sub-transform k1
//   pass B  radix-16 over b in registers; twiddle W64^(a q)
//   pass C  radix-4 over a across the four lanes of a quad (DPP operands of the butterflies' adds, no LDS)
//   result  lane 4 k1 + a, register q holds Z[k1 + 16 q + 256 rev(a)], rev = (0, 2, 1, 3)
// The real-input untangle works on the pairs (k, 1024 - k) that share Z[k] and Z[1024 - k]:
// the partner value is a register of another lane (ds_bpermute, no LDS image of Z); the power spectrum
// is left in LDS for the mel stage.
//
// A wave owns a RUN of consecutive frames of one slice (a whole slice when it has at most 256 frames, as
// the reference's 33..166-frame slices do) and walks it in rounds of 4 frames.  The MFCC rows of its last
// three rounds stay in an LDS ring, from which the same wave forms the 5-tap delta
// (python_speech_features.delta(feat, 2), core/process_data.py:277: the slice's own edge rows replicated) two
// rows behind the MFCCs and writes both halves of a feature row together; a run that ends inside a long
// slice computes two more MFCC rows on either side instead of exchanging them with its neighbour.  Waves
// never wait for each other after the block's set-up.
//
// Occupancy is the design constraint: a wave stalls on its own LDS exchanges, so the kernel keeps three
// waves per SIMD (<= 168 VGPRs, 52 KB of LDS per 4-wave block): the input is read straight from global
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

constexpr int kWavesPerBlock = 4;                            // the block's waves share the DCT table, nothing else
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kRingRows = 3 * kFramesPerWave;                // MFCC rows a wave keeps for the delta stencil
constexpr int kRingBytes = kRingRows * kNCep * 4;
constexpr int kDctBytes = kDctSteps * 64 * 8;                // block-shared B operands of the DCT matrix product
constexpr int kSmemBytes = kWavesPerBlock * (kWaveLdsBytes + kRingBytes) + kDctBytes;
constexpr int kWavesPerSimd = 3;                             // register budget 168; three 52 KB blocks per CU
constexpr int kDeltaReach = 2;                               // delta(feat, 2) looks two rows to either side
constexpr int kMaxRunFrames = 256;                           // frames one wave computes for one run, halo rows included
constexpr int kLongRun = kMaxRunFrames - 2 * kDeltaReach;    // owned frames per run of a slice longer than kMaxRunFrames

// How the frames of a launch are cut into slices, and the slices into runs (one wave per run).
struct SliceMap {
    const int64_t* starts;      // ragged slices of one contiguous stream: n_slices + 1 first-frame indices; or nullptr:
    long frames_per_slice;      //   uniform slices: every slice has this many frames, the last one what is left
    long slice_stride;          //   uniform slices: samples from one slice's alignment rows to the next slice's
    long n_frames;              // frames of the launch (= starts[n_slices])
    long n_runs;                // n_slices * runs_per_slice
    int runs_per_slice;         // waves per slice
    int run_len;                // owned frames per run
};

// Rows [j_first, j_first + n_emit) of a slice of n frames, from the wave's MFCC ring (row q at slot
// (q - first) % kRingRows): 12 coefficients and their deltas, d[j] = (-2 f[j-2] - f[j-1] + f[j+1] + 2 f[j+2]) / 10
// with the slice's own edge rows replicated.  One lane per (row, coefficient): it reads the five ring rows of its
// stencil once and stores the coefficient and its delta (48-byte runs of a packed 24-float row).
__device__ __forceinline__ void emit_rows(const float* ring, int first, int n, int j_first, int n_emit, int lane,
                                          float* __restrict__ mfcc, long mfcc_ld, float* __restrict__ delta, long delta_ld) {
    for (int idx = lane; idx < n_emit * kNCep; idx += 64) {
        const int r = idx / kNCep, c = idx - r * kNCep;
        const int j = j_first + r;
        auto at = [&](int q) {
            q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
            return ring[((q - first) % kRingRows) * kNCep + c];
        };
        const float m2 = at(j - 2), m1 = at(j - 1), f0 = at(j), p1 = at(j + 1), p2 = at(j + 2);
        mfcc[(long)j * mfcc_ld + c] = f0;
        delta[(long)j * delta_ld + c] = (2.0f * (p2 - m2) + (p1 - m1)) / 10.0f;
    }
}

// grid: one wave per run, four waves per block.  Frame j of a slice whose alignment rows start at `rows` reads
// aligned rows j .. j+3 and their predecessors j-1 .. j+2 (row -1 does not exist: zero_first_pred), and is output
// row g_lo + j.  Consecutive frames of a wave re-read three of their four rows: those hits are served by the
// CU's vector L1 / the L2.
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void mfcc_kernel(const int16_t* __restrict__ in,
                                                                 const MfccTables* __restrict__ tab, SliceMap m,
                                                                 float* __restrict__ mfcc, long mfcc_ld,
                                                                 float* __restrict__ delta, long delta_ld) {
    __shared__ __align__(16) unsigned char smem[kSmemBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    unsigned char* wbase = smem + wave * (kWaveLdsBytes + kRingBytes);
    float2* ex = reinterpret_cast<float2*>(wbase);                          // exchange 1 / 2, power spectrum
    float* en = reinterpret_cast<float*>(wbase + kExBytes);                  // [4 frames][32]
    float* parts = reinterpret_cast<float*>(wbase + kPartOffset);            // [2][64], inside the exchange region
    float* ring = reinterpret_cast<float*>(wbase + kWaveLdsBytes);           // [kRingRows][12]
    const double* dctb = reinterpret_cast<const double*>(smem + kWavesPerBlock * (kWaveLdsBytes + kRingBytes));
    const int off512 = tab->pow_off_512;

    LaneConst K;
    load_lane_const(K, tab, lane);
    {
        const double* dsrc = &tab->dctb[0][0];
        double* ddst = reinterpret_cast<double*>(smem + kWavesPerBlock * (kWaveLdsBytes + kRingBytes));
        for (int i = threadIdx.x; i < kDctSteps * 64; i += kThreads) ddst[i] = dsrc[i];
        __syncthreads();
    }

    // wave-uniform from here on (the only block barrier is behind us): the wave's run
    const long run = (long)blockIdx.x * kWavesPerBlock + wave;
    if (run >= m.n_runs) return;
    const long s = run / m.runs_per_slice;
    const int part = (int)(run - s * m.runs_per_slice);
    long g_lo;
    int n;                                                   // frames of the slice
    const int16_t* rows;
    if (m.starts) {
        // the table is read through vector loads: tell the compiler the values are wave-uniform, or every loop
        // variable derived from them occupies a vector register
        // The table is the caller's and is not validated on the host (it lives on the device): both ends are
        // clamped into 0 .. n_frames and a slice is cut at 2^30 frames, so that a bad table can leave rows
        // unwritten or computed with the wrong neighbours but cannot read or write outside the buffers.
        long hi = (long)m.starts[s + 1];
        hi = hi < 0 ? 0 : (hi > m.n_frames ? m.n_frames : hi);
        long lo = (long)m.starts[s];
        lo = lo < 0 ? 0 : (lo > hi ? hi : lo);
        if (hi - lo > (1L << 30)) hi = lo + (1L << 30);
        g_lo = ((long)__builtin_amdgcn_readfirstlane((int)(lo >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(lo & 0xFFFFFFFFL));
        n = __builtin_amdgcn_readfirstlane((int)(hi - lo));
        rows = in + g_lo * kFrame;
    } else {
        g_lo = s * m.frames_per_slice;
        n = (int)(m.n_frames - g_lo < m.frames_per_slice ? m.n_frames - g_lo : m.frames_per_slice);
        rows = in + s * m.slice_stride;
    }
    const int* slice_dwords = reinterpret_cast<const int*>(rows);
    mfcc += g_lo * mfcc_ld;                                  // row j of the slice is output row g_lo + j
    delta += g_lo * delta_ld;

    // a slice longer than runs_per_slice * run_len (ragged slices beyond the caller's stated maximum) wraps around
#pragma unroll 1
    for (long j0l = (long)part * m.run_len; j0l < n; j0l += (long)m.runs_per_slice * m.run_len) {
        const int j0 = (int)j0l;
        const int own_hi = n - j0 < m.run_len ? n : j0 + m.run_len;         // owned frames [j0, own_hi)
        const int first = j0 > kDeltaReach ? j0 - kDeltaReach : 0;          // computed frames [first, last)
        const int last = n - own_hi < kDeltaReach ? n : own_hi + kDeltaReach;
        int emitted = j0;                                                   // rows [j0, emitted) are written
#pragma unroll 1
        for (int fbase = first; fbase < last; fbase += kFramesPerWave) {
            int nfr = 0;                                                    // frames of this round
#pragma unroll 1
            for (int it = 0; it < kFramesPerWave; ++it) {
                const int j = fbase + it;
                if (j >= last) break;
                nfr = it + 1;
                // (fetching the wave's next frame under the FFT instead was measured slower, with and without
                // register spills: 18.3 -> 18.8 ms per 100 h at 130 registers; again with the registers the
                // two-sum mel stage freed, 162 in all: 16.9 -> 17.4)
                RawFrame raw;
                FE_STAMP(0);
                frame_load(raw, slice_dwords + (long)j * (kFrame / 2) + lane, j == 0, lane);
                if (it > 0) filter_sums(parts, en + (it - 1) * 32, K);      // the previous frame's, under this load
                frame_energies(raw, off512, K, ex, parts);
                FE_STAMP(9);
                if (it + 1 < kFramesPerWave && j + 1 < last) FE_NEXT_FRAME();
            }
            filter_sums(parts, en + (nfr - 1) * 32, K);
            wave_lds_fence();
            FE_STAMP(10);
            const int tl = fresh_lane();                                    // see fresh_lane: nothing of the tail's
            cepstra_tail(en, reinterpret_cast<double*>(wbase), dctb, nfr, tl, ring, (fbase - first) % kRingRows, 1,
                         kRingRows);                                        // addressing stays live over the frames
            FE_STAMP(11);
            // MFCC rows [first, avail) are in the ring; a row can be written once the rows its delta reads are:
            // everything owned in the last round, otherwise up to two rows behind.
            const int avail = last - fbase > kFramesPerWave ? fbase + kFramesPerWave : last;
            const int upto = avail == last ? own_hi : (avail - kDeltaReach < own_hi ? avail - kDeltaReach : own_hi);
            const int n_emit = upto > emitted ? upto - emitted : 0;
            emit_rows(ring, first, n, emitted, n_emit, tl, mfcc, mfcc_ld, delta, delta_ld);
            emitted += n_emit;
            wave_lds_fence();                                               // ring reads before the next round's writes
            FE_STAMP(12);
            FE_NEXT_FRAME();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: tables

double hz2mel(double hz) { return 2595.0 * log10(1.0 + hz / 700.0); }
double mel2hz(double mel) { return 700.0 * (pow(10.0, mel / 2595.0) - 1.0); }

void build_tables(MfccTables& t) {
    memset(&t, 0, sizeof(t));
    const double pi = 3.14159265358979323846;
    for (int k1 = 1; k1 < 16; ++k1)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * pi * (double)(l * k1) / 1024.0;
            t.tw1[k1 - 1][l] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int q = 1; q < 16; ++q)
        for (int a4 = 0; a4 < 4; ++a4) {
            const double a = -2.0 * pi * (double)(a4 * q) / 64.0, sgn = a4 ? -1.0 : 1.0;
            t.tw2[q - 1][a4] = make_float2((float)(sgn * cos(a)), (float)(sgn * sin(a)));
        }
    for (int j = 0; j < 8; ++j)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * pi * (double)fft_bin(l, j) / 2048.0;
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
    // where the FFT lanes put their bins: bin k of run-lane L sits at float kRunSlot * L + (k - run_start[L]);
    // bin 1024 belongs to no filter and goes to the last float of the (idle) last lane's slot
    {
        int slot_of[1025];
        for (int k = 0; k <= 1024; ++k) slot_of[k] = kRunSlot * 63 + kRunSlot - 1;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < t.run_count[l]; ++i) slot_of[t.run_start[l] + i] = kRunSlot * l + i;
        if (t.run_count[63] == kRunSlot) abort();
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = fft_bin(l, j);
                t.pow_off[0][l][j] = (unsigned short)(4 * slot_of[k]);
                t.pow_off[1][l][j] = (unsigned short)(4 * slot_of[1024 - k]);
            }
        t.pow_off_512 = 4 * slot_of[512];
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
    for (int s = 0; s < kDctSteps; ++s)
        for (int l = 0; l < 64; ++l) {
            const int n = 4 * s + (l >> 4), k = l & 15;
            t.dctb[s][l] = (n < kNFilt && k < kNCep) ? t.dct[n][k] : 0.0;
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

namespace {

// One launch of the front end over the slices described by `m` (runs_per_slice and run_len are set here).
int launch_mfcc(const int16_t* frames_dev, SliceMap m, long n_slices, long max_slice_frames, float* mfcc_dev, long mfcc_ld,
                float* delta_dev, long delta_ld, hipStream_t st) {
    int rc = ensure_mfcc_tables();
    if (rc != VAD_OK) return rc;
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    if (max_slice_frames <= kMaxRunFrames) {                      // a block per slice: no halo rows anywhere
        m.run_len = (int)max_slice_frames;
        m.runs_per_slice = 1;
    } else {
        m.run_len = kLongRun;
        const long runs = (max_slice_frames + kLongRun - 1) / kLongRun;
        VAD_REQUIRE(runs <= 0x7FFFFFFFL, "slice too long for one launch");
        m.runs_per_slice = (int)runs;
    }
    m.n_runs = n_slices * m.runs_per_slice;
    const long blocks = (m.n_runs + kWavesPerBlock - 1) / kWavesPerBlock;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many frames for one launch");
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, frames_dev, g_tables[dev], m, mfcc_dev,
                       mfcc_ld, delta_dev, delta_ld);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

}  // namespace
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
    SliceMap m{};
    m.starts = nullptr;
    m.frames_per_slice = (long)frames_per_slice;
    m.slice_stride = (long)slice_stride;
    m.n_frames = (long)((n_slices - 1) * frames_per_slice + last_slice_frames);
    return launch_mfcc(frames_dev, m, (long)n_slices, (long)frames_per_slice, mfcc_dev, (long)mfcc_ld, delta_dev,
                       (long)delta_ld, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int vad_mfcc_delta_slices_f32(const int16_t* frames_dev, const int64_t* slice_starts_dev, int64_t n_slices,
                                         int64_t n_frames, int64_t max_slice_frames, float* mfcc_dev, int64_t mfcc_ld,
                                         float* delta_dev, int64_t delta_ld, void* stream) {
    using namespace vad;
    VAD_REQUIRE(frames_dev && slice_starts_dev && mfcc_dev && delta_dev, "null pointer argument");
    VAD_REQUIRE(n_slices >= 1 && n_frames >= 1, "n_slices and n_frames must be >= 1");
    VAD_REQUIRE(max_slice_frames >= 1, "max_slice_frames must be >= 1");
    VAD_REQUIRE(mfcc_ld >= kNCep && delta_ld >= kNCep, "leading dimensions must be >= 12");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(frames_dev) & 3) == 0, "frames_dev must be 4-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(slice_starts_dev) & 7) == 0, "slice_starts_dev must be 8-byte aligned");
    VAD_REQUIRE(n_slices < (int64_t(1) << 31) && n_frames < (int64_t(1) << 40), "size out of range");
    SliceMap m{};
    m.starts = slice_starts_dev;
    m.n_frames = (long)n_frames;
    return launch_mfcc(frames_dev, m, (long)n_slices, (long)max_slice_frames, mfcc_dev, (long)mfcc_ld, delta_dev,
                       (long)delta_ld, reinterpret_cast<hipStream_t>(stream));
}

#if defined(VAD_FE_TIMELINE)
extern "C" int vad_debug_fe_timeline(unsigned long long* out_host) {
    VAD_HIP_CHECK(hipDeviceSynchronize());
    VAD_HIP_CHECK(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vad::fe::g_fe_timeline), sizeof(vad::fe::g_fe_timeline)));
    const int zero = 0;
    VAD_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vad::fe::g_fe_tl_frame), &zero, sizeof(int)));
    return vad::fe::kFeTlFrames * vad::fe::kFeTlStamps;
}
#endif
