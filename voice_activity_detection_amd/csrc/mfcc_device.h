// Device-side building blocks of the MFCC front end (kernel and launch code: mfcc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "vad_internal.h"

namespace vad {
namespace fe {

constexpr int kFrame = VAD_FRAME_SIZE;        // 480
constexpr int kNFilt = 26;
constexpr int kNCep = VAD_NUM_CEPS;           // 12
constexpr int kFramesPerWave = 4;              // frames whose log/DCT tail a wave batches
constexpr int kMaxRun = 20;                   // bins per lane in the mel stage
constexpr int kDctSteps = 7;                  // k-steps of 4 covering the 26 filters in the DCT matrix product
constexpr int kExStride = 68;                 // the exchange: complex elements per k1 row (64 + 4: the eight k1 rows that
                                              // the lanes of a 32-lane half read land 8 banks apart)
constexpr int kExBytes = 16 * kExStride * 8;  // 8704
constexpr int kRunSlot = 20;                  // power spectrum in LDS: the run of mel lane l occupies floats [20 l, 20 l + 20),
                                              // so the lanes' 16-byte reads of their runs fall on distinct banks
constexpr int kPowFloats = 64 * kRunSlot;
constexpr float kPowScale = 0.25f / 2048.0f;  // 2^-13

struct MfccTables {
    float2 tw1[15][64];      // [k1-1][lane]          W1024^(lane * k1)
    float2 tw2[15][4];       // [q-1][a]              s_a W64^(a * q), s = (1, -1, -1, -1): the signs pass C expects
    float2 twr[8][64];       // [u][lane]             W2048^k, k = fft_bin(lane, u)
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
    double dctb[kDctSteps][64];  // the same matrix as B operands of v_mfma_f64_16x16x4_f64: step s, lane l holds
                                 // dct[4 s + (l >> 4)][l & 15], zero outside 26 x 12
    unsigned short pow_off[2][64][8];   // byte offset, inside the power-spectrum image, of the bins FFT lane l produces:
                                        // [0][l][u] bin k = fft_bin(l, u), [1][l][u] bin 1024 - k
    int pow_off_512;                    // the same for bin 512 (lane 1's self-paired bin)
};

// After pass C lane l = 4 k1 + a holds Z[k1 + 16 q + 256 rev(a)] in register q (rev = 0, 2, 1, 3).  The untangle
// pairs it with Z[1024 - k]; slot u of a lane is the pair it works on in step u: register u and its partner
// (register 15 - u of lane 4 (16 - k1) + 3 - a; for k1 = 0 register 16 - u of lane 3 - a).  Step 0 of lanes 0..3
// takes what that rule leaves over: lane 0 bins 0 / 1024 (register 0, self-paired), lane 2 bins 256 / 768 (register
// 0, partner lane 3), lanes 1 and 3 register 8 (bins 640 / 384 with lane 2, bins 896 / 128 with lane 0).
__host__ __device__ inline int fft_bin(int lane, int u) {
    const int k1 = lane >> 2, a = lane & 3, rev = ((a & 1) << 1) | (a >> 1);
    if (k1 == 0 && u == 0 && (a & 1)) return 128 + 256 * rev;
    return k1 + 16 * u + 256 * rev;
}

static_assert(offsetof(MfccTables, pow_off) % 16 == 0 && offsetof(MfccTables, dctb) % 8 == 0,
              "tables that are copied to LDS in 16-byte pieces");

// ---------------------------------------------------------------------------------------------
// device helpers

// Diagnostic build only (-DVAD_FE_TIMELINE=<block>): s_memtime stamps of wave 0 of one block at the stage boundaries
// of its first frames, read back through vad_debug_fe_timeline (tools/timeline_fe.py).  A stamp waits for the
// wave's outstanding LDS and scalar operations, so the stages it separates do not overlap as they do in the product.
#if defined(VAD_FE_TIMELINE)
constexpr int kFeTlFrames = 96, kFeTlStamps = 16;
__device__ unsigned long long g_fe_timeline[kFeTlFrames][kFeTlStamps];
__device__ int g_fe_tl_frame;
#define FE_STAMP(k)                                                                                           \
    do {                                                                                                      \
        const unsigned long long fe_now = __builtin_amdgcn_s_memtime();                                       \
        if (blockIdx.x == VAD_FE_TIMELINE && threadIdx.x == 0 && g_fe_tl_frame < kFeTlFrames)                 \
            g_fe_timeline[g_fe_tl_frame][k] = fe_now;                                                         \
    } while (0)
#define FE_NEXT_FRAME()                                                                                       \
    do {                                                                                                      \
        if (blockIdx.x == VAD_FE_TIMELINE && threadIdx.x == 0) g_fe_tl_frame = g_fe_tl_frame + 1;             \
    } while (0)
#else
#define FE_STAMP(k)
#define FE_NEXT_FRAME()
#endif

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

// The lane index, recomputed where it is called.  Addresses derived from a lane index that the compiler can
// see is loop-invariant are hoisted to kernel entry and kept in vector registers across the whole frame loop
// (or spilled around it); two instructions per use are cheaper than those registers.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// LDS traffic below is between lanes of ONE wave.  LDS operations of a wave
// complete in issue order, so only the compiler has to be kept from moving
// loads above the stores of other lanes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Per-wave LDS: [the exchange, then power spectrum | 2 x 64 mel partial sums, then log energies][filter energies
// of 4 frames].  The partial sums of a frame sit behind its power spectrum and are summed while the next frame
// loads, before that frame's exchange overwrites them.
constexpr int kEnBytes = kFramesPerWave * 32 * 4;
constexpr int kPartBytes = 2 * 64 * 4;
constexpr int kPartOffset = kPowFloats * 4;
constexpr int kWaveLdsBytes = kExBytes + kEnBytes;
static_assert(kPartOffset + kPartBytes <= kExBytes, "the power spectrum and the partial sums must fit the exchange region");

// Loop-invariant per-lane constants that stay in registers.
struct LaneConst {
    float2 tw1[15];                           // W1024^(lane k1)                 } all three twiddle sets: 76 of the
    float2 tw2[15];                           // s_a W64^(a q) of this lane's a  } 168 registers three waves per SIMD
    float2 twr[8];                            // W2048^fft_bin(lane, u)          } leave a wave
    unsigned pow_lo[8], pow_hi[8];            // byte offsets of the lane's bins in the power-spectrum image
    int run_count;
    float run_inv, run_w0, run_f0;
    int rise_first, rise_n, fall_first, fall_n;
};

__device__ __forceinline__ void load_lane_const(LaneConst& K, const MfccTables* __restrict__ tab, int lane) {
#pragma unroll
    for (int k = 0; k < 15; ++k) K.tw2[k] = tab->tw2[k][lane & 3];
#pragma unroll
    for (int k = 0; k < 15; ++k) K.tw1[k] = tab->tw1[k][lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) K.twr[j] = tab->twr[j][lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        K.pow_lo[j] = tab->pow_off[0][lane][j];
        K.pow_hi[j] = tab->pow_off[1][lane][j];
    }
    const int run_start = tab->run_start[lane];
    K.run_count = tab->run_count[lane];
    const float b0 = (float)tab->run_b0[lane], b1 = (float)tab->run_b1[lane];
    // The power spectrum is left unscaled in LDS, 4 nfft |X|^2 / nfft = 2^13 times psf's |X|^2 / nfft; the mel
    // weights carry the 2^-13 instead (a power of two: the same bits as scaling every bin, one multiply less per bin)
    const float inv = tab->run_inv[lane];
    K.run_inv = inv * kPowScale;
    K.run_w0 = ((float)run_start - b0) * inv * kPowScale;  // rising weight of the run's first bin
    K.run_f0 = (b1 - (float)run_start) * inv * kPowScale;  // falling weight of the run's first bin
    const int fl = lane < kNFilt ? lane : 0;
    K.rise_first = tab->rise_first[fl];
    K.rise_n = tab->rise_n[fl];
    K.fall_first = tab->fall_first[fl];
    K.fall_n = tab->fall_n[fl];
}

// Raw input of one frame as this lane holds it: int16 pairs n = lane + 64 jj of the frame's four rows
// (cur) and of the rows one frame earlier (prv, the pre-emphasis predecessors).
struct RawFrame {
    int cur[15], prv[15];
};

// `cur_p` points (in global memory) at this lane's first int16 pair of the frame's first row; the frame's
// four rows are contiguous and the row 240 dwords earlier is the pre-emphasis predecessor.  With
// zero_first_pred the predecessor of the frame's FIRST row is taken as zero and not read (first frame of a
// slice: that row is not pre-emphasised, as in the reference's 2-D preemphasis).  Neighbouring frames
// re-read the same rows; those hits are served by the CU's vector L1 / the L2.
__device__ __forceinline__ void frame_load(RawFrame& raw, const int* __restrict__ cur_p, bool zero_first_pred, int lane) {
#pragma unroll
    for (int jj = 0; jj < 15; ++jj) {
        raw.cur[jj] = cur_p[64 * jj];
        raw.prv[jj] = 0;
        if (!(jj < 4 && zero_first_pred && (lane + 64 * jj) < kFrame / 2)) raw.prv[jj] = cur_p[64 * jj - kFrame / 2];
    }
}

// One frame from its raw input to the mel partial sums of the 64 run-lanes in parts[0..63] (rising edges) and
// parts[64..127] (falling edges); filter_sums turns them into the 26 filter energies.
// The W1024 twiddle factors live in 30 registers (K.tw1).  They used to be read from a block-shared LDS image while the
// mel stage still needed the registers; with 37 freed there, taking the 15 reads per frame off the LDS array is worth
// 16.4 -> 15.9 ms per 100 h (reading them from global memory makes the kernel texture-path-bound: 2.2x slower).
// `off512` is MfccTables::pow_off_512.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void frame_energies(const RawFrame& raw, int off512, const LaneConst& K,
                                               float2* ex, float* parts) {
    const int lane = fresh_lane();
    const int k1l = lane >> 2, al = lane & 3;
    float* pw = reinterpret_cast<float*>(ex);                               // aliases ex once the exchange is read

    // ---- pre-emphasis: z[n] = y[2n] + i y[2n+1], n = lane + 64 jj ----
    float2 v[16];
#pragma unroll
    for (int jj = 0; jj < 15; ++jj) {
        const int cur = raw.cur[jj], prv = raw.prv[jj];
        const float x0 = fmaf(-0.97f, (float)(short)(prv & 0xFFFF), (float)(short)(cur & 0xFFFF));
        const float x1 = fmaf(-0.97f, (float)(prv >> 16), (float)(cur >> 16));
        v[jj] = make_float2(x0, x1);
    }
    v[15] = make_float2(0.0f, 0.0f);                        // samples 1920..2047 are zero padding
#if defined(VAD_FE_TIMELINE)
    asm volatile("" : "+v"(v[0].x), "+v"(v[14].y));          // the samples have arrived
#endif
    FE_STAMP(1);

    // ---- pass A: radix 16 over jj in registers; register k1 then holds sub-transform k1 of point n2 = lane;
    //      twiddle W1024^(n2 k1) ----
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], K.tw1[k - 1]);

    // ---- the exchange: [k1][n2] -> lane (k1, a) reads n2 = a + 4 b, b = 0..15 ----
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) ex[k1 * kExStride + lane] = v[k1];
    wave_lds_fence();
    FE_STAMP(2);
    {
        const float2* rd = ex + k1l * kExStride + al;
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = rd[4 * b];
    }
    wave_lds_fence();
    FE_STAMP(3);

    // ---- pass B: radix 16 over b -> q; twiddle W64^(a q), with the signs pass C expects (tw2) ----
    dft16(v);
    {
        const float s0 = al ? -1.0f : 1.0f;
        v[0] = make_float2(v[0].x * s0, v[0].y * s0);
    }
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], K.tw2[k - 1]);
#if defined(VAD_FE_TIMELINE)
    asm volatile("" : "+v"(v[1].x), "+v"(v[15].y));
#endif
    FE_STAMP(4);

    // ---- pass C: radix 4 over a, across the four lanes of a quad (DPP operands, no LDS).  With x_a the
    //      twiddled values (lanes 1..3 hold them negated):
    //        stage 1 (partner a ^ 2)   r = x + c1 P:   lane 0 x0 + x2, lane 1 -(x1 + x3), lane 2 x0 - x2, lane 3 x1 - x3
    //        lane 3 turns its value by i
    //        stage 2 (partner a ^ 1)   X = r + c2 P:   lane 0 X0, lane 1 X2, lane 2 X1, lane 3 X3
    //      so lane (k1, a) ends up with Z[k1 + 16 q + 256 rev(a)] in register q. ----
    {
        const float c1 = ((al + 1) & 2) ? 1.0f : -1.0f;     // -1, 1, 1, -1
        const float c2 = (al & 1) ? 1.0f : -1.0f;           // -1, 1, -1, 1
        const bool turn = al == 3;
        // r += c * (r of the quad partner): one v_fmac_f32 whose first operand is fetched through DPP.  Eight at a
        // time in one asm statement: the two wait states a DPP read needs after a vector write of the same
        // register are paid once per statement (the compiler does not track hazards into inline asm), and the
        // eight instructions touch eight different registers.
#define VAD_QUAD_FMAC8(CTRL, R)                                                                                  \
        asm volatile("s_nop 1\n\t"                                                                                \
                     "v_fmac_f32_dpp %0, %0, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %1, %1, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %2, %2, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %3, %3, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %4, %4, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %5, %5, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %6, %6, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_fmac_f32_dpp %7, %7, %8 " CTRL " row_mask:0xf bank_mask:0xf"                                \
                     : "+v"(R[0].x), "+v"(R[0].y), "+v"(R[1].x), "+v"(R[1].y), "+v"(R[2].x), "+v"(R[2].y),       \
                       "+v"(R[3].x), "+v"(R[3].y)                                                                \
                     : "v"(cc))
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float2* r = v + 4 * g;
            float cc = c1;
            VAD_QUAD_FMAC8("quad_perm:[2,3,0,1]", r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float rx = r[q].x, ry = r[q].y;
                r[q] = make_float2(turn ? -ry : rx, turn ? rx : ry);
            }
            cc = c2;
            VAD_QUAD_FMAC8("quad_perm:[1,0,3,2]", r);
        }
#undef VAD_QUAD_FMAC8
    }
    FE_STAMP(5);

    // ---- real-input untangle on the pairs (k, 1024 - k) (fft_bin: which pair a lane takes in step u):
    //      with a = Z[k], b = Z[1024 - k], E = a + conj(b), O = a - conj(b), T = W2048^k O,
    //      4 |X[k]|^2 = |E - iT|^2 and 4 |X[1024 - k]|^2 = |E + iT|^2.
    //      b is a register of another lane (a cross-lane read, no LDS image of Z): the SENDER picks the register
    //      its reader wants in step u. ----
    const bool zl = k1l == 0;                                                // lanes 0..3: the bins with k1 = 0
    const int plane = zl ? 3 - al : 4 * (16 - k1l) + 3 - al;
    const int plane0 = zl ? (((al + 1) & 2) ? al + 1 : 0) : plane;            // step 0 of lanes 0..3: 0, 2, 3, 0
    const int paddr = plane * 4, paddr0 = plane0 * 4;
    unsigned char* pwb = reinterpret_cast<unsigned char*>(pw);
    // all sixteen cross-lane reads are issued ahead of their use: their latency is paid once
    float2 partner[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float2 send;
        if (u == 0) {
            const float2 s0 = al == 3 ? v[0] : v[8];
            send = zl ? s0 : v[15];
        } else {
            send = zl ? v[16 - u] : v[15 - u];
        }
        partner[u].x = __int_as_float(__builtin_amdgcn_ds_bpermute(u ? paddr : paddr0, __float_as_int(send.x)));
        partner[u].y = __int_as_float(__builtin_amdgcn_ds_bpermute(u ? paddr : paddr0, __float_as_int(send.y)));
    }
    __builtin_amdgcn_sched_barrier(0);
    FE_STAMP(6);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        float2 a = v[jj];
        float2 b = partner[jj];
        if (jj == 0) {
            if (zl && (al & 1)) a = v[8];
            if (lane == 0) b = v[0];
        }
        const float2 e = make_float2(a.x + b.x, a.y - b.y);
        const float2 o = make_float2(a.x - b.x, a.y + b.y);
        const float2 t = cmul(K.twr[jj], o);
        const float ar = e.x + t.y, ai = e.y - t.x;                       // E - iT
        const float br = e.x - t.y, bi = e.y + t.x;                       // E + iT
        const unsigned o_lo = K.pow_lo[jj], o_hi = K.pow_hi[jj];
        *reinterpret_cast<float*>(pwb + o_lo) = ar * ar + ai * ai;                            // bin k (x 2^13)
        *reinterpret_cast<float*>(pwb + o_hi) = br * br + bi * bi;                            // bin 1024 - k
    }
    if (lane == 1)                                                                             // the self-paired bin 512
        *reinterpret_cast<float*>(pwb + off512) = (v[0].x * v[0].x + v[0].y * v[0].y) * 4.0f;
    wave_lds_fence();
    FE_STAMP(7);

    // ---- mel segments: each lane sums a run of <= 20 bins of one segment (no branches: the lane reads
    // its whole 20-float slot in five 16-byte reads, elements past its run count as zeros) ----
    {
        float sr = 0.0f, sf = 0.0f;
        static_assert(kMaxRun == kRunSlot && kRunSlot % 4 == 0, "a run is read as whole 16-byte pieces of its slot");
        const f32x4v* run4 = reinterpret_cast<const f32x4v*>(pw + kRunSlot * lane);
        f32x4v piece[kRunSlot / 4];
#pragma unroll
        for (int q = 0; q < kRunSlot / 4; ++q) piece[q] = run4[q];
        // The weights of a run are affine in the bin's position i: rising w0 + i inv, falling f0 - i inv.  Two plain
        // sums over the run, S0 = sum p_i and S1 = sum i p_i (all terms non-negative), then
        // rising = w0 S0 + inv S1, falling = f0 S0 - inv S1: three vector instructions per bin instead of five.
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < kMaxRun; ++i) {
            const float p = (i < K.run_count) ? piece[i >> 2][i & 3] : 0.0f;
            s0 += p;
            if (i) s1 = fmaf((float)i, p, s1);
        }
        sr = fmaf(K.run_inv, s1, K.run_w0 * s0);
        sf = fmaf(-K.run_inv, s1, K.run_f0 * s0);
        parts[lane] = sr;
        parts[64 + lane] = sf;
    }
    FE_STAMP(8);
}

// filter f = rising parts of segment f + falling parts of segment f+1, in a fixed order; en_slot[0..25].
// Called after the NEXT frame's samples have been requested (or before the tail): the LDS round trip of the
// partial sums passes under that wait instead of stalling the wave at the end of every frame.
__device__ __forceinline__ void filter_sums(const float* parts, float* en_slot, const LaneConst& K) {
    wave_lds_fence();
    const int lane = fresh_lane();
    float e = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) e += (i < K.rise_n) ? parts[min(K.rise_first + i, 63)] : 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) e += (i < K.fall_n) ? parts[64 + min(K.fall_first + i, 63)] : 0.0f;
    if (lane < 32) en_slot[lane] = e;
}

// A double-precision constant the compiler cannot see through: the tail below sits inside the kernel's frame
// loops, and constants it could name would be hoisted out of them into vector registers that the FFT needs
// (the hoisted copies were spilled to scratch memory and reloaded one dependent load after the other).
template <unsigned HI, unsigned LO>
__device__ __forceinline__ double opaque_f64() {
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"(LO), "i"(HI));
    return __hiloint2double((int)hi, (int)lo);
}

// Natural logarithm of a positive, normal double (every positive float32 widens to one), absolute error
// below 3e-14 over the float32 range: x = m 2^e with m in [sqrt(1/2), sqrt 2), s = (m - 1) / (m + 1),
// log m = 2 s + 2 s^3 / 3 + ... + 2 s^15 / 15.
__device__ __forceinline__ double log_pos(double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    int e = ((hi >> 20) & 0x7FF) - 1023;
    double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, lo);               // [1, 2)
    const bool big = m > opaque_f64<0x3FF6A09Eu, 0x667F3BCDu>();                    // sqrt 2
    m = big ? m * 0.5 : m;
    e += big ? 1 : 0;
    const double d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double t = (m - 1.0) * r;
    const double z = t * t;
    double p = opaque_f64<0x3FC11111u, 0x11111111u>();                             // 2/15
    p = fma(p, z, opaque_f64<0x3FC3B13Bu, 0x13B13B14u>());                         // 2/13
    p = fma(p, z, opaque_f64<0x3FC745D1u, 0x745D1746u>());                         // 2/11
    p = fma(p, z, opaque_f64<0x3FCC71C7u, 0x1C71C71Cu>());                         // 2/9
    p = fma(p, z, opaque_f64<0x3FD24924u, 0x92492492u>());                         // 2/7
    p = fma(p, z, opaque_f64<0x3FD99999u, 0x9999999Au>());                         // 2/5
    p = fma(p, z, opaque_f64<0x3FE55555u, 0x55555555u>());                         // 2/3
    return fma((double)e, opaque_f64<0x3FE62E42u, 0xFEFA39EFu>(), fma(t * z, p, t + t));   // e ln 2 + log m
}

// Tail for up to 4 frames of one wave at once: log of the 26 energies (fp64), then the lifted DCT-II as one
// small fp64 matrix product on the matrix cores, C[frame][k] = sum_n log E[frame][n] * dct[n][k] with
// v_mfma_f64_16x16x4_f64 (M = 16 rows of which 4 are frames, N = 16 columns of which 12 are coefficients,
// K = 26 in 7 steps of 4): lane l supplies A[l & 15][4 s + (l >> 4)] and B[4 s + (l >> 4)][l & 15] and ends up
// with C[4 r + (l >> 4)][l & 15] in accumulator element r (layout measured by tools/probes/mfma_f64_layout.hip),
// so element 0 of lane l is coefficient l & 15 of frame l >> 4.
// The 12 coefficients of frame `it` go to rows[slot(it) * 12 ..), slot(it) = (slot0 + it * slot_step) % n_slots,
// a ring of MFCC rows in LDS.  `logv` may alias the exchange region; `dctb` is the block's LDS copy of MfccTables::dctb.
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cepstra_tail(const float* en, double* logv, const double* __restrict__ dctb, int nfr, int lane,
                                             float* rows, int slot0, int slot_step, int n_slots) {
#pragma unroll
    for (int r = 0; r < kFramesPerWave / 2; ++r) {
        const int it = 2 * r + (lane >> 5), f = lane & 31;
        if (f < kNFilt && it < nfr) {
            const float e = en[it * 32 + f];
            logv[it * kNFilt + f] = log_pos(e == 0.0f ? opaque_f64<0x3CB00000u, 0u>() : (double)e);   // psf.fbank clamp: eps = 2^-52
        }
    }
    wave_lds_fence();
    {
        const int i = lane & 15, kk = lane >> 4;
        double b[kDctSteps], a[kDctSteps];
#pragma unroll
        for (int s = 0; s < kDctSteps; ++s) b[s] = dctb[s * 64 + lane];
#pragma unroll
        for (int s = 0; s < kDctSteps; ++s) {
            const int n = 4 * s + kk;
            a[s] = (i < nfr && n < kNFilt) ? logv[i * kNFilt + n] : 0.0;     // rows of frames that do not exist: 0
        }
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < kDctSteps; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
        if (i < kNCep && kk < nfr) rows[((slot0 + kk * slot_step) % n_slots) * kNCep + i] = (float)c[0];
    }
    wave_lds_fence();
}

}  // namespace fe

// Device copy of the constant tables for the current device (created on first use); mfcc.hip.
const fe::MfccTables* mfcc_tables_for_current_device();

}  // namespace vad
