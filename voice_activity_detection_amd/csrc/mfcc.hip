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
// Z is stored transposed ([q2][lane], a linear image) and the real-input untangle works on the
// pairs (k, 1024 - k), k < 512, that share Z[k] and Z[1024 - k]; the power spectrum is left in LDS
// for the mel stage.
//
// Precision: FFT, power and mel sums in fp32; filter energies, log and DCT in
// fp64 (26 logs and 312 fused multiply-adds per frame), because an fp32 log of
// an energy near e^25 already carries 2e-6 absolute error before the DCT.

#include <math.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "vad_internal.h"

namespace vad {

namespace {

constexpr int kFrame = VAD_FRAME_SIZE;        // 480
constexpr int kNFilt = 26;
constexpr int kNCep = VAD_NUM_CEPS;           // 12
constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerWave = 4;
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;
constexpr int kMaxRun = 20;                   // bins per lane in the mel stage
constexpr int kEx1Stride = 272;               // exchange 1: complex elements per k1 row (256 + 16: the two
                                              // k1 rows of a 32-lane half land on opposite bank halves)
constexpr int kEx2Plane = 273;                // exchange 2: complex elements per k1 plane, 17 per q1 row
constexpr int kExBytes = 8736;                // >= 4*272*8, >= (3*273 + 15*17 + 16)*8, >= 1024*8
constexpr int kPowFloats = 1056;              // P[0..1024] plus slack for reads past a lane's run

struct MfccTables {
    float2 tw1[12][64];      // [(k1-1)*4 + b][lane]  W1024^((lane + 64 b) * k1)
    float2 tw2[15][16];      // [q1-1][c]             W256^(c * q1)
    float2 twr[8][64];       // [jj][lane]            W2048^(k1 + 4 q1 + 64 jj), lane = 16 k1 + q1
    int run_start[64];       // mel stage: first bin of the lane's run
    int run_count[64];       //            bins in the run (0 for idle lanes)
    int run_b0[64];          //            first bin of the run's segment
    int run_b1[64];          //            one past the last bin of the segment
    float run_inv[64];       //            1 / (b1 - b0)
    int rise_first[kNFilt];  // filter f = rising parts of lanes [rise_first, +rise_n)
    int rise_n[kNFilt];
    int fall_first[kNFilt];  //          + falling parts of lanes [fall_first, +fall_n)
    int fall_n[kNFilt];
    double dct[kNFilt][kNCep];   // sqrt(2/26) * lifter(k) * cos(pi k (2n+1) / 52), k = 1..12
};

// ---------------------------------------------------------------------------------------------
// device helpers

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// Forward 4-point DFT in place (W4 = -i).
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3);
    const float2 d = csub(a1, a3);
    const float2 t3 = make_float2(d.y, -d.x);          // (a1 - a3) * (-i)
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

// Forward 16-point DFT in place, natural order in and out.
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;      // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;      // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // v[4c + b] *= W16^(b c)
    v[5] = cmul(v[5], make_float2(C1, -S1));           // b=1 c=1
    v[9] = cmul(v[9], make_float2(R2, -R2));           // b=1 c=2
    v[13] = cmul(v[13], make_float2(S1, -C1));         // b=1 c=3
    v[6] = cmul(v[6], make_float2(R2, -R2));           // b=2 c=1
    v[10] = make_float2(v[10].y, -v[10].x);            // b=2 c=2  (-i)
    v[14] = cmul(v[14], make_float2(-R2, -R2));        // b=2 c=3
    v[7] = cmul(v[7], make_float2(S1, -C1));           // b=3 c=1
    v[11] = cmul(v[11], make_float2(-R2, -R2));        // b=3 c=2
    v[15] = cmul(v[15], make_float2(-C1, S1));         // b=3 c=3  W16^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // X[c + 4 d] sits at v[4 c + d]: transpose the 4x4 register block.
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c + 1; d < 4; ++d) {
            const float2 t = v[4 * c + d];
            v[4 * c + d] = v[4 * d + c];
            v[4 * d + c] = t;
        }
}

// LDS traffic below is between lanes of ONE wave.  LDS operations of a wave
// complete in issue order, so only the compiler has to be kept from moving
// loads above the stores of other lanes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kStageRows = kFramesPerBlock + 4;              // predecessor row + 16 frames + 3 trailing rows
constexpr int kStageDwords = kStageRows * (kFrame / 2);       // int16 pairs
constexpr int kWaveLdsBytes = kExBytes /*exchanges, Z, partial sums, log energies*/ +
                              kPowFloats * 4 /*power spectrum*/ + kFramesPerWave * 32 * 4 /*filter energies*/;
constexpr int kSmemBytes = kStageDwords * 4 + kNFilt * kNCep * 8 + kWavesPerBlock * kWaveLdsBytes;

// grid: blocks_per_slice blocks for each slice; a block owns 16 consecutive frames of ONE slice, so
// its 20 input rows are contiguous whatever slice_stride is.  Wave w takes frames w, w+4, w+8, w+12.
__global__ __launch_bounds__(256, 2) void mfcc_kernel(const int16_t* __restrict__ in, const MfccTables* __restrict__ tab,
                                                      long n_slices, long frames_per_slice, long last_slice_frames,
                                                      long slice_stride, int blocks_per_slice,
                                                      float* __restrict__ out, long out_ld) {
    __shared__ __align__(16) unsigned char smem[kSmemBytes];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    int* stage = reinterpret_cast<int*>(smem);
    double* dct_s = reinterpret_cast<double*>(smem + kStageDwords * 4);
    unsigned char* wbase = smem + kStageDwords * 4 + kNFilt * kNCep * 8 + wave * kWaveLdsBytes;
    float2* ex = reinterpret_cast<float2*>(wbase);                          // exchange 1 / 2, then Zt[16][64]
    float* part_r = reinterpret_cast<float*>(wbase);                        // aliases ex once Z is consumed
    float* part_f = part_r + 64;
    double* logv = reinterpret_cast<double*>(wbase);                        // aliases ex in the tail
    float* pw = reinterpret_cast<float*>(wbase + kExBytes);                  // power spectrum
    float* en = pw + kPowFloats;                                             // [4 frames][32]

    const long s = blockIdx.x / blocks_per_slice;
    const int j0 = (int)(blockIdx.x - s * blocks_per_slice) * kFramesPerBlock;
    const long n_s = (s == n_slices - 1) ? last_slice_frames : frames_per_slice;
    if (j0 >= n_s) return;                                                   // block-uniform

    // ---- stage aligned rows j0-1 .. j0+18 of the slice (zeros outside it) ----
    {
        const int16_t* slice = in + s * slice_stride;
        const long rows = n_s + VAD_ALIGN_FRAMES;
        if ((reinterpret_cast<uintptr_t>(slice) & 15) == 0) {
            const uint4* src = reinterpret_cast<const uint4*>(slice);
            uint4* dst = reinterpret_cast<uint4*>(stage);
            for (int i = threadIdx.x; i < kStageDwords / 4; i += 256) {
                const int row = i / 60, c4 = i - row * 60;
                const long ar = (long)j0 - 1 + row;
                dst[i] = (ar >= 0 && ar < rows) ? src[ar * 60 + c4] : make_uint4(0, 0, 0, 0);
            }
        } else {
            const int* src = reinterpret_cast<const int*>(slice);
            for (int i = threadIdx.x; i < kStageDwords; i += 256) {
                const int row = i / 240, c = i - row * 240;
                const long ar = (long)j0 - 1 + row;
                stage[i] = (ar >= 0 && ar < rows) ? src[ar * 240 + c] : 0;
            }
        }
    }
    for (int i = threadIdx.x; i < kNFilt * kNCep; i += 256) dct_s[i] = (&tab->dct[0][0])[i];

    // Loop-invariant per-lane constants.
    float2 tw1[12], tw2[15], twr[8];
#pragma unroll
    for (int k = 0; k < 12; ++k) tw1[k] = tab->tw1[k][lane];
#pragma unroll
    for (int k = 0; k < 15; ++k) tw2[k] = tab->tw2[k][lane & 15];
#pragma unroll
    for (int j = 0; j < 8; ++j) twr[j] = tab->twr[j][lane];
    const int run_start = tab->run_start[lane];
    const int run_count = tab->run_count[lane];
    const float run_b0 = (float)tab->run_b0[lane];
    const float run_b1 = (float)tab->run_b1[lane];
    const float run_inv = tab->run_inv[lane];
    const int fl = lane < kNFilt ? lane : 0;
    const int rise_first = tab->rise_first[fl], rise_n = tab->rise_n[fl];
    const int fall_first = tab->fall_first[fl], fall_n = tab->fall_n[fl];

    const float run_w0 = ((float)run_start - run_b0) * run_inv;      // rising weight of the run's first bin
    const float run_f0 = (run_b1 - (float)run_start) * run_inv;      // falling weight of the run's first bin
    // untangle partner of bin k = k1 + 4 q1 + 64 jj: Zt[(15 - jj + pd) & 15][plane]
    const int k1l = lane >> 4, q1l = lane & 15;
    const int plane = k1l ? 16 * (4 - k1l) + 15 - q1l : (q1l ? 16 - q1l : 0);
    const int pd = (lane == 0) ? 1 : 0;
    const int kbase = k1l + 4 * q1l;

    __syncthreads();

    int nfr = 0;                                            // frames this wave produced (wave-uniform)
#pragma unroll 1
    for (int it = 0; it < kFramesPerWave; ++it) {
        const int jl = wave + it * kWavesPerBlock;          // frame index inside the block
        if (j0 + jl >= n_s) break;
        nfr = it + 1;

        // ---- load + pre-emphasis: z[n] = y[2n] + i y[2n+1], n = lane + 64 jj.  Frame jl covers staged
        // rows jl+1 .. jl+4; the sample one row (480 samples) earlier is 240 dwords back.  The row in
        // front of a slice's first aligned row is staged as zeros, which leaves that row un-emphasised
        // exactly as the reference's 2-D preemphasis does. ----
        float2 v[16];
        {
            const int* cur_p = stage + (jl + 1) * (kFrame / 2) + lane;
#pragma unroll
            for (int jj = 0; jj < 15; ++jj) {
                const int cur = cur_p[64 * jj];
                const int prv = cur_p[64 * jj - kFrame / 2];
                const float x0 = fmaf(-0.97f, (float)(short)(prv & 0xFFFF), (float)(short)(cur & 0xFFFF));
                const float x1 = fmaf(-0.97f, (float)(prv >> 16), (float)(cur >> 16));
                v[jj] = make_float2(x0, x1);
            }
            v[15] = make_float2(0.0f, 0.0f);                // samples 1920..2047 are zero padding
        }

        // ---- pass A: radix 4 over n1 = jj >> 2; outputs k1 land in v[4 k1 + b] ----
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
#pragma unroll
        for (int k = 0; k < 12; ++k) v[4 + k] = cmul(v[4 + k], tw1[k]);

        // ---- exchange 1: [k1][n2 = lane + 64 b] -> lane (k1, c) reads n2 = c + 16 m ----
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int b = 0; b < 4; ++b) ex[k1 * kEx1Stride + lane + 64 * b] = v[4 * k1 + b];
        wave_lds_fence();
        {
            const float2* rd = ex + k1l * kEx1Stride + q1l;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = rd[16 * m];
        }
        wave_lds_fence();

        // ---- pass B: radix 16 over m; twiddle W256^(c q1) ----
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw2[k - 1]);

        // ---- exchange 2: [k1][q1][c] -> lane (k1, q1) reads c = 0..15 ----
        {
            float2* wr = ex + k1l * kEx2Plane + q1l;
#pragma unroll
            for (int q = 0; q < 16; ++q) wr[17 * q] = v[q];
        }
        wave_lds_fence();
        {
            const float2* rd = ex + k1l * kEx2Plane + 17 * q1l;
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = rd[c];
        }
        wave_lds_fence();

        // ---- pass C: radix 16 over c; register q2 now holds Z[k1 + 4 q1 + 64 q2] ----
        dft16(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) ex[64 * q + lane] = v[q];             // Zt[q2][lane]
        wave_lds_fence();

        // ---- real-input untangle on the pairs (k, 1024 - k), k = k1 + 4 q1 + 64 jj < 512:
        //      with E = Z[k] + conj(Z[N-k]), O = Z[k] - conj(Z[N-k]), T = W2048^k O,
        //      4 |X[k]|^2 = |E - iT|^2 and 4 |X[N-k]|^2 = |E + iT|^2 ----
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float2 a = ex[64 * jj + lane];
            const float2 b = ex[64 * ((15 - jj + pd) & 15) + plane];
            const float2 e = make_float2(a.x + b.x, a.y - b.y);
            const float2 o = make_float2(a.x - b.x, a.y + b.y);
            const float2 t = cmul(twr[jj], o);
            const float ar = e.x + t.y, ai = e.y - t.x;                   // E - iT
            const float br = e.x - t.y, bi = e.y + t.x;                   // E + iT
            const int k = kbase + 64 * jj;
            pw[k] = (ar * ar + ai * ai) * (0.25f / 2048.0f);
            pw[1024 - k] = (br * br + bi * bi) * (0.25f / 2048.0f);
        }
        if (lane == 0) {                                                  // the self-paired bin 512
            const float2 a = ex[64 * 8];
            pw[512] = (a.x * a.x + a.y * a.y) * (1.0f / 2048.0f);
        }
        wave_lds_fence();

        // ---- mel segments: each lane sums a run of <= 20 bins of one segment (no branches: the
        // 20 reads issue back to back, lanes past their run add zeros) ----
        {
            float sr = 0.0f, sf = 0.0f;
            const float* run = pw + run_start;
#pragma unroll
            for (int i = 0; i < kMaxRun; ++i) {
                const float p = (i < run_count) ? run[i] : 0.0f;
                sr = fmaf(p, fmaf((float)i, run_inv, run_w0), sr);
                sf = fmaf(p, fmaf(-(float)i, run_inv, run_f0), sf);
            }
            part_r[lane] = sr;                                            // ex is free again (Z consumed)
            part_f[lane] = sf;
        }
        wave_lds_fence();

        // ---- filter f = rising parts of segment f + falling parts of segment f+1, fixed order ----
        {
            float e = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; ++i) e += (i < rise_n) ? part_r[min(rise_first + i, 63)] : 0.0f;
#pragma unroll
            for (int i = 0; i < 5; ++i) e += (i < fall_n) ? part_f[min(fall_first + i, 63)] : 0.0f;
            if (lane < 32) en[it * 32 + lane] = e;
        }
        wave_lds_fence();
    }

    // ---- tail for the wave's (up to) 4 frames at once: log of the 26 energies, lifted DCT-II ----
#pragma unroll
    for (int r = 0; r < kFramesPerWave / 2; ++r) {
        const int it = 2 * r + (lane >> 5), f = lane & 31;
        if (f < kNFilt && it < nfr) {
            const float e = en[it * 32 + f];
            logv[it * kNFilt + f] = log(e == 0.0f ? 2.220446049250313e-16 : (double)e);   // psf.fbank clamp
        }
    }
    wave_lds_fence();
    if (lane < kFramesPerWave * kNCep) {
        const int it = lane / kNCep, k = lane - it * kNCep;
        if (it < nfr) {
            double c = 0.0;
#pragma unroll
            for (int n = 0; n < kNFilt; ++n) c = fma(logv[it * kNFilt + n], dct_s[n * kNCep + k], c);
            const long g = s * frames_per_slice + j0 + wave + it * kWavesPerBlock;
            out[g * out_ld + k] = (float)c;
        }
    }
}

// delta[g] = (-2 f[g-2] - f[g-1] + f[g+1] + 2 f[g+2]) / 10 with the slice's own edge rows
// replicated (python_speech_features.delta(feat, 2), core/process_data.py:277).
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ mfcc, long mfcc_ld, long n_frames,
                                                    long frames_per_slice, float* __restrict__ delta, long delta_ld) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * kNCep) return;
    const long g = idx / kNCep;
    const int c = (int)(idx - g * kNCep);
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
    const long bps = (frames_per_slice + kFramesPerBlock - 1) / kFramesPerBlock;
    const long blocks = bps * n_slices;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many frames for one launch");
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)blocks), dim3(256), 0, st, frames_dev, g_tables[dev], (long)n_slices,
                       (long)frames_per_slice, (long)last_slice_frames, (long)slice_stride, (int)bps, mfcc_dev,
                       (long)mfcc_ld);
    VAD_HIP_CHECK(hipGetLastError());
    const long dthreads = n_frames * kNCep;
    const long dblocks = (dthreads + 255) / 256;
    VAD_REQUIRE(dblocks <= 0x7FFFFFFFL, "too many frames for one launch");
    hipLaunchKernelGGL(delta_kernel, dim3((unsigned)dblocks), dim3(256), 0, st, mfcc_dev, (long)mfcc_ld, n_frames,
                       (long)frames_per_slice, delta_dev, (long)delta_ld);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}
