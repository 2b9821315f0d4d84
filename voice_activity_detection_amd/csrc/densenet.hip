// DenseNet window classifier for gfx950: DenseNet.forward of the reference
// (core/models.py:383-461) in eval mode, BatchNorm folded into the convolutions.
//
// One wave owns 32 windows.  Its activations live in LDS as act[channel][position][window]
// (window = lane & 31, so every access is bank-conflict free).
//
//  * Wide contractions -- the dilated stem convolution (K = 24 x 6 = 144) and the two 1x1
//    convolutions (K = 42 or 80) -- run on the fp32 matrix cores exactly like the recurrent
//    model: out^T (channel x window) = W (channel x k) . in^T (k x window) with
//    v_mfma_f32_32x32x2_f32; the result has the window on the lane and 16 output channels in the
//    accumulator registers, and each lane stores its channels straight to LDS.
//  * The growth convolutions (3 or 4 output channels, kernel 3) would fill a tenth of a 32-row
//    MFMA tile, so they run on the vector ALU: the two lane halves split the input channels,
//    accumulate growth x positions partial sums each and combine them with one cross-half
//    exchange per output.
//
// Order inside every stage is the reference's: convolution -> BatchNorm -> MaxPool(2) ->
// LeakyReLU(0.01) (core/models.py:391-398); Dropout is the identity in eval mode.

#include <math.h>
#include <string.h>
#include <mutex>

#include "vad_internal.h"

namespace vad {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int kF = VAD_FEATURES;    // 24 input channels
constexpr int kTaps = 6;            // stem kernel size, dilation 4
constexpr int kL1 = 5;              // positions inside dense block 1 (30 -> 10 -> pool 5)
constexpr int kL2 = 2;              // positions inside dense block 2 (5 -> pool 2)

template <bool LARGE>
struct Dims {
    static constexpr int c0 = LARGE ? 48 : 24;       // stem / transition channels
    static constexpr int nl = LARGE ? 8 : 6;         // layers per dense block
    static constexpr int g = LARGE ? 4 : 3;          // growth rate
    static constexpr int c1 = c0 + nl * g;           // channels after a dense block (80 / 42)
    static constexpr int mt0 = (c0 + 31) / 32;       // 32-row tiles covering c0
    static constexpr int mt1 = (c1 + 31) / 32;       // 32-row tiles covering c1
    static constexpr int ks1 = c1 / 2;               // k-steps of a 1x1 convolution over c1 channels
    // Growth channels kept in registers instead of LDS (dense_block): the small model all of them (63 registers),
    // the large model those of block 1's last three layers and all of block 2's (more would not fit 256 registers).
    static constexpr int reg1 = LARGE ? nl - 3 : 0;  // first register-resident growth layer of dense block 1
    static constexpr int reg2 = 0;                   //                                         of dense block 2
    // private LDS floats of a wave's 32 windows: block 1's LDS-resident growth channels, then block 2's inputs
    static constexpr int priv1 = reg1 * g * 5 * 32, priv2 = (c0 + reg2 * g) * 2 * 32;
    static constexpr int priv = priv1 > priv2 ? priv1 : priv2;
};

// Blob layout in floats.
struct DnLayout {
    int a_in, s_in;              // stem: A [mt0][6][12][64], shift [mt0][2][16]
    int dense[2];                // dense block b: per layer [C_in][12] weights (o * 3 + tap, zero padded) + shift [4]
    int a_tr, s_tr;              // transition: A [mt0][c1/2][64], shift [mt0][2][16]
    int a_out, s_out;            // cnn_out:    A [mt1][c1/2][64], shift [mt1][2][16]
    int w_fin;                   // out.weight: [mt1][2 classes][2 halves][16]
    int total;
};

// Offset of growth layer i inside its dense block: layers 0..i-1 hold (c0 + g k) * 12 + 4 floats each.
__host__ __device__ constexpr int dense_layer_offset(int c0, int g, int i) {
    return i * (c0 * 12 + 4) + 6 * g * i * (i - 1);
}

template <bool LARGE>
__host__ __device__ inline DnLayout dn_layout() {
    using D = Dims<LARGE>;
    DnLayout L;
    int o = 0;
    L.a_in = o; o += D::mt0 * kTaps * 12 * 64;
    L.s_in = o; o += D::mt0 * 32;
    for (int b = 0; b < 2; ++b) {
        L.dense[b] = o;
        o += dense_layer_offset(D::c0, D::g, D::nl);
    }
    L.a_tr = o; o += D::mt0 * D::ks1 * 64;
    L.s_tr = o; o += D::mt0 * 32;
    L.a_out = o; o += D::mt1 * D::ks1 * 64;
    L.s_out = o; o += D::mt1 * 32;
    L.w_fin = o; o += D::mt1 * 2 * 2 * 16;
    L.total = o;
    return L;
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : 0.01f * v; }

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x of this lane + x of the lane 32 away, in every lane: v_permlane32_swap exchanges the upper half of one
// register with the lower half of another in the vector ALU (ds_bpermute would go through the LDS pipe).
__device__ __forceinline__ float sum_across_halves(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = lower half twice, r[1] = upper half twice
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ f32x16 load_shift(const float* p, int half) {
    const float4* s = reinterpret_cast<const float4*>(p + half * 16);
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = s[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
    return r;
}

// Where a wave's activations live.  Channels below `split` (the stem / transition outputs) are read at
// a[c * a_c + l * a_l], the growth channels appended by a dense block at b[(c - split) * b_c + l * b_l];
// both bases already include the lane's window.  The stride-1 kernel points `a` at a block-shared image of
// the stem output, the generic kernel at its private copy.
struct ActView {
    const float* a;
    int a_c, a_l;
    float* b;
    int b_c, b_l;
    int split;
    __device__ __forceinline__ float ld(int c, int l) const {
        return c < split ? a[c * a_c + l * a_l] : b[(c - split) * b_c + l * b_l];
    }
};

// One dense block: NL growth layers over L positions, appended channel by channel.  The outputs of layers
// REG_FROM.. stay in registers instead of LDS: the lane halves split the input channels by parity, so a growth
// channel is only ever read by one half, and gr[s][l] holds relative channel G REG_FROM + 2 s + half at position l
// (G REG_FROM even).  With REG_FROM = NL everything goes to LDS.
//
// STAGED: the weights (in global memory) reach the FMAs through `ring`, 256 floats of LDS private to the wave.
// Every lane of a wave needs the same twelve weights per channel; read straight from global memory that is three
// 16-byte loads per channel whose ~1 000 cycles of latency two waves per SIMD cannot cover (and there are no
// registers to request more channels ahead).  Instead one coalesced 16-byte load per LANE fetches a chunk of 20
// channels (+ the 16 floats behind them: the layer's shift row when the chunk is its last) a whole chunk ahead of
// its use, four registers in flight; when the channel loop reaches the chunk it is written to the ring and read
// back as broadcasts.  The values and the order of the arithmetic are those of the direct reads.
constexpr int kStageChannels = 20;
constexpr int kStageFloats = 256;

template <int NL, int G, int L, int REG_FROM, bool STAGED = false>
__device__ __forceinline__ void dense_block(const ActView& v, const float* block_w, int c_first, int half,
                                            float (&gr)[((NL - REG_FROM) * G + 1) / 2 + 1][L], float* ring = nullptr) {
    static_assert((G * REG_FROM) % 2 == 0, "the register-resident channels start at an even channel");
    static_assert(!STAGED || G % 2 == 0, "staged weights: every layer starts on an even channel");
    constexpr int kRegFirst = G * REG_FROM;                       // first relative channel kept in registers
    int ring_base = 0;                                            // first channel of the chunk in the ring (wave-uniform)
    float4 pre = make_float4(0.0f, 0.0f, 0.0f, 0.0f);             // the lane's 16 bytes of the chunk after that
    int stage_lane = 0;
    if constexpr (STAGED) {
        stage_lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        pre = *reinterpret_cast<const float4*>(block_w + 4 * stage_lane);
    }
    if constexpr (REG_FROM < NL) {
#pragma unroll
        for (int s2 = 0; s2 < ((NL - REG_FROM) * G + 1) / 2; ++s2)
#pragma unroll
            for (int l = 0; l < L; ++l) gr[s2][l] = 0.0f;
    }
#pragma unroll
    for (int layer = 0; layer < NL; ++layer) {                    // unrolled: the channel count below is a constant
        const int cin = c_first + G * layer;
        const float* wl = block_w + dense_layer_offset(c_first, G, layer);
        // (staged) the chunk that starts at channel `base` of this layer moves from `pre` into the ring, and the
        // chunk after it -- the next layer's first one behind this layer's last -- is requested
        auto install = [&](int base) {
            wave_lds_fence();                                     // the reads of the chunk it replaces
            reinterpret_cast<float4*>(ring)[stage_lane] = pre;
            wave_lds_fence();
            ring_base = base;
            const float* nxt = base + kStageChannels < cin
                                   ? wl + (base + kStageChannels) * 12
                                   : block_w + dense_layer_offset(c_first, G, layer + 1 < NL ? layer + 1 : layer);
            pre = *reinterpret_cast<const float4*>(nxt + 4 * stage_lane);
        };
        float acc[G][L];
#pragma unroll
        for (int o = 0; o < G; ++o)
#pragma unroll
            for (int l = 0; l < L; ++l) acc[o][l] = 0.0f;
        auto accumulate = [&](int c, const float (&a)[L]) {
            const float4* wp = STAGED ? reinterpret_cast<const float4*>(ring + (c - ring_base) * 12)
                                      : reinterpret_cast<const float4*>(wl + c * 12);
            const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
            const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int o = 0; o < G; ++o)
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    float s = fmaf(w[o * 3 + 1], a[l], acc[o][l]);
                    if (l > 0) s = fmaf(w[o * 3 + 0], a[l - 1], s);          // padding = 1
                    if (l < L - 1) s = fmaf(w[o * 3 + 2], a[l + 1], s);
                    acc[o][l] = s;
                }
        };
        auto channel = [&](int c, const float* src, int sl) {
            float a[L];
#pragma unroll
            for (int l = 0; l < L; ++l) a[l] = src[l * sl];
            accumulate(c, a);
        };
        // The lane halves take the even / the odd channels (c_first is even, so both halves leave the first
        // range together).  The weight loads of several channels have to be in flight at once: the unrolled
        // iterations are independent until the accumulation.
        const int lds_end = layer < REG_FROM ? cin : c_first + kRegFirst;   // growth channels that live in LDS
        const int n_reg = (REG_FROM < NL && layer > REG_FROM) ? (layer - REG_FROM) * G : 0;   // register-resident inputs
        // the channels [lo, hi) of this layer, in order: stem / transition outputs, growth channels in LDS, in registers
        auto channels = [&](int lo, int hi) {
            const int hi_a = hi < c_first ? hi : c_first;
#pragma unroll 4
            for (int c = lo + half; c < hi_a; c += 2) channel(c, v.a + c * v.a_c, v.a_l);
            const int lo_b = lo > c_first ? lo : c_first, hi_b = hi < lds_end ? hi : lds_end;
#pragma unroll 4
            for (int c = lo_b + half; c < hi_b; c += 2) channel(c, v.b + (c - c_first) * v.b_c, v.b_l);
            if constexpr (REG_FROM < NL) {
#pragma unroll
                for (int s2 = 0; s2 < (n_reg + 1) / 2; ++s2) {
                    const int ce = c_first + kRegFirst + 2 * s2;
                    if (ce < lo || ce >= hi) continue;
                    if (2 * s2 + 1 < n_reg) {
                        accumulate(ce + half, gr[s2]);
                    } else {
                        // the odd channel of the last pair does not exist yet: the upper half multiplies zeros (by
                        // the rows that follow in the weight table: finite values)
                        float a[L];
#pragma unroll
                        for (int l = 0; l < L; ++l) a[l] = half ? 0.0f : gr[s2][l];
                        accumulate(ce + half, a);
                    }
                }
            }
        };
        if constexpr (STAGED) {
#pragma unroll
            for (int base = 0; base < cin; base += kStageChannels) {
                install(base);
                channels(base, base + kStageChannels < cin ? base + kStageChannels : cin);
            }
        } else {
            channels(0, cin);
        }
        const float4 sh4 = STAGED ? *reinterpret_cast<const float4*>(ring + (cin - ring_base) * 12)
                                  : *reinterpret_cast<const float4*>(wl + cin * 12);
        const float sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        if (layer < REG_FROM) {
#pragma unroll
            for (int o = 0; o < G; ++o)
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const float y = lrelu(sum_across_halves(acc[o][l]) + sh[o]);
                    if (half == 0) v.b[(cin + o - c_first) * v.b_c + l * v.b_l] = y;
                }
            wave_lds_fence();
        } else if constexpr (REG_FROM < NL) {
#pragma unroll
            for (int o = 0; o < G; ++o) {
                const int rel = (layer - REG_FROM) * G + o;      // relative to the first register-resident channel
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const float y = lrelu(sum_across_halves(acc[o][l]) + sh[o]);
                    gr[rel / 2][l] = (half == (rel & 1)) ? y : gr[rel / 2][l];
                }
            }
        }
    }
}

// Stem for one tile of 32 columns: Conv1d(24 -> c0, k = 6, dilation 4) + BN at two adjacent positions
// (`xe` and the row after it), max-pooled and passed through LeakyReLU.  out(mt, i, value) receives
// accumulator register i of channel tile mt.
template <bool LARGE, typename Out>
__device__ __forceinline__ void stem_pair(const float* xe_row, const float* __restrict__ blob, const DnLayout& LY, int lane,
                                          int half, Out out) {
    using D = Dims<LARGE>;
    f32x16 acc_e[D::mt0], acc_o[D::mt0];
#pragma unroll
    for (int mt = 0; mt < D::mt0; ++mt) acc_e[mt] = acc_o[mt] = load_shift(blob + LY.s_in + mt * 32, half);
#pragma unroll
    for (int j = 0; j < kTaps; ++j) {
        const float* re = xe_row + 4 * j * kF;                   // tap j reads the row 4 j later (dilation 4)
        const float4 e0 = *reinterpret_cast<const float4*>(re), e1 = *reinterpret_cast<const float4*>(re + 4),
                     e2 = *reinterpret_cast<const float4*>(re + 8);
        const float4 o0 = *reinterpret_cast<const float4*>(re + kF), o1 = *reinterpret_cast<const float4*>(re + kF + 4),
                     o2 = *reinterpret_cast<const float4*>(re + kF + 8);
        const float xe[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
        const float xo[12] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w, o2.x, o2.y, o2.z, o2.w};
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int mt = 0; mt < D::mt0; ++mt) {
                const float a = blob[LY.a_in + ((mt * kTaps + j) * 12 + s) * 64 + lane];
                acc_e[mt] = MFMA(a, xe[s], acc_e[mt]);
                acc_o[mt] = MFMA(a, xo[s], acc_o[mt]);
            }
    }
#pragma unroll
    for (int mt = 0; mt < D::mt0; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) out(mt, i, lrelu(fmaxf(acc_e[mt][i], acc_o[mt][i])));
}

// Stem convolution + BN at ONE position per column: acc[mt] = S[row] for the lane's channels, the value stem_pair
// computes as its "even" half (the same fmaf chain).
template <bool LARGE>
__device__ __forceinline__ void stem_single(const float* x_row, const float* __restrict__ blob, const DnLayout& LY, int lane,
                                            int half, f32x16 (&acc)[Dims<LARGE>::mt0]) {
    using D = Dims<LARGE>;
#pragma unroll
    for (int mt = 0; mt < D::mt0; ++mt) acc[mt] = load_shift(blob + LY.s_in + mt * 32, half);
#pragma unroll
    for (int j = 0; j < kTaps; ++j) {
        const float* re = x_row + 4 * j * kF;                    // tap j reads the row 4 j later (dilation 4)
        const float4 e0 = *reinterpret_cast<const float4*>(re), e1 = *reinterpret_cast<const float4*>(re + 4),
                     e2 = *reinterpret_cast<const float4*>(re + 8);
        const float xe[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int mt = 0; mt < D::mt0; ++mt)
                acc[mt] = MFMA(blob[LY.a_in + ((mt * kTaps + j) * 12 + s) * 64 + lane], xe[s], acc[mt]);
    }
}

// Everything after the stem for a wave's 32 windows: dense block 1 over `v1`, transition into the plain
// private array act2[c][2][32] (which may alias v1's growth region), dense block 2, cnn_out, Linear, softmax.
// `dense_w[b]` points at the growth-layer weights of dense block b (the blob's, or a copy in LDS).
// REG1 / REG2: first growth layer of dense block 1 / 2 whose outputs stay in registers (D::nl = none).
// STAGED / ring: see dense_block.
template <bool LARGE, int REG1 = Dims<LARGE>::nl, int REG2 = Dims<LARGE>::nl, bool STAGED = false>
__device__ __forceinline__ void densenet_tail(const ActView& v1, float* act2, const float* __restrict__ blob,
                                              const float* dense_w0, const float* dense_w1,
                                              const DnLayout& LY, int lane, long w, long n_windows,
                                              float* __restrict__ p_out, uint8_t* __restrict__ dec_out, int mode,
                                              float threshold, float* ring = nullptr) {
    using D = Dims<LARGE>;
    const int half = lane >> 5, col = lane & 31;

    // ---- dense block 1 (5 positions) ----
    float gr1[((D::nl - REG1) * D::g + 1) / 2 + 1][kL1];
    dense_block<D::nl, D::g, kL1, REG1, STAGED>(v1, dense_w0, D::c0, half, gr1, ring);
    constexpr int kSReg1 = (D::c0 + D::g * REG1) / 2;           // first k-step of the 1x1 convolution fed from gr1

    // ---- transition: 1x1 conv c1 -> c0 + BN + MaxPool(2) + LeakyReLU, 5 -> 2 (position 4 is dropped) ----
    f32x16 tr[D::mt0][kL2];
#pragma unroll
    for (int lp = 0; lp < kL2; ++lp) {
        f32x16 acc_e[D::mt0], acc_o[D::mt0];
#pragma unroll
        for (int mt = 0; mt < D::mt0; ++mt) acc_e[mt] = acc_o[mt] = load_shift(blob + LY.s_tr + mt * 32, half);
#pragma unroll 8
        for (int s = 0; s < kSReg1; ++s) {
            const int c = 2 * s + half;
            const float be = v1.ld(c, 2 * lp);
            const float bo = v1.ld(c, 2 * lp + 1);
#pragma unroll
            for (int mt = 0; mt < D::mt0; ++mt) {
                const float a = blob[LY.a_tr + (mt * D::ks1 + s) * 64 + lane];
                acc_e[mt] = MFMA(a, be, acc_e[mt]);
                acc_o[mt] = MFMA(a, bo, acc_o[mt]);
            }
        }
#pragma unroll
        for (int s = kSReg1; s < D::ks1; ++s) {                  // channels 2 s + half live in registers
#pragma unroll
            for (int mt = 0; mt < D::mt0; ++mt) {
                const float a = blob[LY.a_tr + (mt * D::ks1 + s) * 64 + lane];
                acc_e[mt] = MFMA(a, gr1[s - kSReg1][2 * lp], acc_e[mt]);
                acc_o[mt] = MFMA(a, gr1[s - kSReg1][2 * lp + 1], acc_o[mt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < D::mt0; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) tr[mt][lp][i] = lrelu(fmaxf(acc_e[mt][i], acc_o[mt][i]));
    }
    wave_lds_fence();                                           // all reads of block 1 are done
#pragma unroll
    for (int lp = 0; lp < kL2; ++lp)
#pragma unroll
        for (int mt = 0; mt < D::mt0; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * mt + mfma32_row(i, half);
                if (c < D::c0) act2[(c * kL2 + lp) * 32 + col] = tr[mt][lp][i];
            }
    wave_lds_fence();

    // ---- dense block 2 (2 positions) ----
    const ActView v2{act2 + col, kL2 * 32, 32, act2 + D::c0 * kL2 * 32 + col, kL2 * 32, 32, D::c0};
    float gr2[((D::nl - REG2) * D::g + 1) / 2 + 1][kL2];
    dense_block<D::nl, D::g, kL2, REG2, STAGED>(v2, dense_w1, D::c0, half, gr2, ring);
    constexpr int kSReg2 = (D::c0 + D::g * REG2) / 2;

    // ---- cnn_out: 1x1 conv c1 -> c1 + BN + MaxPool(2) + LeakyReLU, 2 -> 1; then Linear(c1 -> 2) ----
    float l0 = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int mt = 0; mt < D::mt1; ++mt) {
        f32x16 acc_e, acc_o;
        acc_e = acc_o = load_shift(blob + LY.s_out + mt * 32, half);
#pragma unroll 8
        for (int s = 0; s < kSReg2; ++s) {
            const int c = 2 * s + half;
            const float a = blob[LY.a_out + (mt * D::ks1 + s) * 64 + lane];
            acc_e = MFMA(a, act2[(c * kL2 + 0) * 32 + col], acc_e);
            acc_o = MFMA(a, act2[(c * kL2 + 1) * 32 + col], acc_o);
        }
#pragma unroll
        for (int s = kSReg2; s < D::ks1; ++s) {
            const float a = blob[LY.a_out + (mt * D::ks1 + s) * 64 + lane];
            acc_e = MFMA(a, gr2[s - kSReg2][0], acc_e);
            acc_o = MFMA(a, gr2[s - kSReg2][1], acc_o);
        }
        const float* wf0 = blob + LY.w_fin + ((mt * 2 + 0) * 2 + half) * 16;
        const float* wf1 = blob + LY.w_fin + ((mt * 2 + 1) * 2 + half) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float y = lrelu(fmaxf(acc_e[i], acc_o[i]));    // padded channels carry zero weights below
            l0 = fmaf(y, wf0[i], l0);
            l1 = fmaf(y, wf1[i], l1);
        }
    }
    l0 = sum_across_halves(l0);
    l1 = sum_across_halves(l1);
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    if (half == 0 && w < n_windows) {
        *reinterpret_cast<float2*>(p_out + 2 * w) = make_float2(p0, p1);
        if (dec_out) dec_out[w] = (mode == VAD_DECIDE_THRESHOLD) ? (p1 >= threshold) : (p1 > p0);
    }
}

// Any window placement: one wave = one block = 32 windows, everything private.
template <bool LARGE>
__global__ __launch_bounds__(64) void densenet_kernel(const float* __restrict__ x, long n_windows, WindowMap wm,
                                                      const float* __restrict__ blob, float* __restrict__ p_out,
                                                      uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using D = Dims<LARGE>;
    // the stem output [c0][5][32], then block 1's LDS-resident growth channels; block 2's inputs reuse the start
    constexpr int kStem = D::c0 * kL1 * 32;
    static_assert(D::priv2 <= kStem + D::priv1, "block 2's inputs fit the region block 1 leaves");
    __shared__ __align__(16) float act[kStem + D::priv1];
    const DnLayout LY = dn_layout<LARGE>();
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    const int col = lane & 31;
    const long w = (long)blockIdx.x * 32 + col;
    const long wc = w < n_windows ? w : n_windows - 1;
    const float* xw = window_ptr(x, wm, wc) + 12 * half;

    // ---- stem: 30 rows -> 10 positions -> pooled to 5 ----
    for (int lp = 0; lp < kL1; ++lp)
        stem_pair<LARGE>(xw + 2 * lp * kF, blob, LY, lane, half, [&](int mt, int i, float y) {
            const int c = 32 * mt + mfma32_row(i, half);
            if (c < D::c0) act[(c * kL1 + lp) * 32 + col] = y;
        });
    wave_lds_fence();

    const ActView v1{act + col, kL1 * 32, 32, act + D::c0 * kL1 * 32 + col, kL1 * 32, 32, D::c0};
    densenet_tail<LARGE, D::reg1, D::reg2>(v1, act, blob, blob + LY.dense[0], blob + LY.dense[1], LY, lane, w, n_windows,
                                           p_out, dec_out, mode, threshold);
}

// Stride-1 windows over a packed (F, 24) matrix.  The stem output at pooled position lp of window w is
// M[w + 2 lp], M[r] = LeakyReLU(max(S[r], S[r + 1])) with S[r] the stem convolution at absolute row r:
// it belongs to the row, not to the window, and ten overlapping windows would each recompute it.  A block
// of NW = 8 waves computes M once for its NW * 32 rows (one 32-row tile per wave) into a shared LDS image and
// then scores its NW * 32 - 9 windows from it; only the growth channels are private to a wave.
template <bool LARGE>
struct DnStreamCfg {
    using D = Dims<LARGE>;
    static constexpr int waves = LARGE ? 8 : 12;
    static constexpr int rows = waves * 32;                      // M rows per block
    static constexpr int windows = rows - 9;                     // window w reads M[w .. w + 8], M[r] needs S[r + 1]
    // 6 KB (small) / 12.8 KB (large) of private LDS per wave: 12 waves = three per SIMD (small), 8 waves = two per
    // SIMD (large) share a CU
    static constexpr int reg1 = D::reg1, reg2 = D::reg2, priv = D::priv;
    // The growth-layer weights of both dense blocks are read once per channel by every wave, two addresses
    // per wave-instruction: the small model keeps a block-shared copy in LDS; the large model's LDS has no room
    // for one (47 KB), so its waves stage the weights chunk by chunk through 1 KB each (dense_block, STAGED).
    static constexpr bool dense_in_lds = !LARGE;
    static constexpr bool dense_staged = LARGE;
    // (+ 16 floats: a layer whose register-resident inputs end on an even channel reads one row past its weights)
    static constexpr int dense_floats = dense_in_lds ? 2 * dense_layer_offset(D::c0, D::g, D::nl) + 16 : waves * kStageFloats;
    static_assert(waves * priv >= D::c0 * rows, "the S image of phase 1 fits the private regions it aliases");
    static constexpr int lds_floats = D::c0 * rows + waves * priv + dense_floats;
};

template <bool LARGE>
__global__ __launch_bounds__(DnStreamCfg<LARGE>::waves * 64) void densenet_stream_kernel(
    const float* __restrict__ feat, long n_windows, const float* __restrict__ blob, float* __restrict__ p_out,
    uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using D = Dims<LARGE>;
    using C = DnStreamCfg<LARGE>;
    extern __shared__ __align__(16) float lds[];
    float* M = lds;                                              // [c0][rows]
    const DnLayout LY = dn_layout<LARGE>();
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wave = threadIdx.x >> 6;
    float* priv = lds + D::c0 * C::rows + wave * C::priv;
    float* dense_lds = lds + D::c0 * C::rows + C::waves * C::priv;
    if (C::dense_in_lds) {                                       // visible after the barrier that ends phase 1
        const float4* src = reinterpret_cast<const float4*>(blob + LY.dense[0]);
        float4* dst = reinterpret_cast<float4*>(dense_lds);
        for (int i = threadIdx.x; i < C::dense_floats / 4; i += C::waves * 64) dst[i] = src[i];
    }
    const long w0 = (long)blockIdx.x * C::windows;
    const long last_row = n_windows + VAD_FRAMES - 2;            // the matrix has n_windows + 29 rows

    // ---- phase 1: S[r], the stem convolution at row r = w0 + 32 wave + col, once per row (a window-by-window
    // evaluation computes every S twice, as the second half of one pooled pair and the first of the next); tap 5
    // reads row r + 20, rows past the matrix (only reached by entries that no valid window uses) are clamped to its
    // last rows.  The S image aliases the waves' private regions, unused until phase 2. ----
    float* S = lds + D::c0 * C::rows;                            // [c0][rows]
    f32x16 sv[D::mt0];
    {
        const long r = w0 + wave * 32 + col;
        const long rc = r + 20 <= last_row ? r : (last_row - 20 > 0 ? last_row - 20 : 0);
        stem_single<LARGE>(feat + rc * kF + 12 * half, blob, LY, lane, half, sv);
#pragma unroll
        for (int mt = 0; mt < D::mt0; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * mt + mfma32_row(i, half);
                if (c < D::c0) S[c * C::rows + wave * 32 + col] = sv[mt][i];
            }
    }
    __syncthreads();
    // M[r] = LeakyReLU(max(S[r], S[r + 1])): the lane's own S[r] and its neighbour's S[r + 1] (the block's last row
    // has no neighbour: its M entry is not used by the block's windows)
    {
        const int rl = wave * 32 + col;
        const int rn = rl + 1 < C::rows ? rl + 1 : rl;
#pragma unroll
        for (int mt = 0; mt < D::mt0; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * mt + mfma32_row(i, half);
                if (c < D::c0) M[c * C::rows + rl] = lrelu(fmaxf(sv[mt][i], S[c * C::rows + rn]));
            }
    }
    __syncthreads();

    // ---- phase 2: the wave's 32 windows (the last wave of a block owns 23) ----
    const int wl = wave * 32 + col;                              // window index inside the block
    const bool owned = wl < C::windows;
    const long w = owned ? w0 + wl : n_windows;                  // not owned: masked at the store
    if (w0 + wave * 32 >= n_windows) return;                     // wave-uniform, after the only barrier
    const int ml = owned ? wl : C::windows - 1;                  // keep M reads inside the image
    const ActView v1{M + ml, C::rows, 2, priv + col, kL1 * 32, 32, D::c0};
    const float* dw0 = C::dense_in_lds ? dense_lds : blob + LY.dense[0];
    const float* dw1 = C::dense_in_lds ? dense_lds + dense_layer_offset(D::c0, D::g, D::nl) : blob + LY.dense[1];
    densenet_tail<LARGE, C::reg1, C::reg2, C::dense_staged>(v1, priv, blob, dw0, dw1, LY, lane, w, n_windows, p_out, dec_out, mode,
                                                            threshold, dense_lds + wave * kStageFloats);
}

// ---------------------------------------------------------------------------------------------
// host: BatchNorm folding and packing

struct ConvBn {          // one Conv1d(bias=False) + BatchNorm1d pair of the state dict
    const float *w, *gamma, *beta, *mean, *var;
};

template <bool LARGE>
int pack(const float* const* t, float* blob) {
    using D = Dims<LARGE>;
    const DnLayout L = dn_layout<LARGE>();
    memset(blob, 0, sizeof(float) * (size_t)L.total);
    int idx = 0;
    auto next = [&]() {
        ConvBn c{t[idx], t[idx + 1], t[idx + 2], t[idx + 3], t[idx + 4]};
        idx += 5;
        return c;
    };
    auto scale = [](const ConvBn& c, int o) { return (double)c.gamma[o] / sqrt((double)c.var[o] + 1e-5); };
    auto shift = [&](const ConvBn& c, int o) { return (double)c.beta[o] - (double)c.mean[o] * scale(c, o); };
    auto put_shift_tiles = [&](const ConvBn& c, int n_out, int n_tiles, float* dst) {
        for (int mt = 0; mt < n_tiles; ++mt)
            for (int hh = 0; hh < 2; ++hh)
                for (int i = 0; i < 16; ++i) {
                    const int o = 32 * mt + mfma32_row(i, hh);
                    if (o < n_out) dst[mt * 32 + hh * 16 + i] = (float)shift(c, o);
                }
    };

    // stem: weight (c0, 24, 6)
    ConvBn cin = next();
    for (int mt = 0; mt < D::mt0; ++mt)
        for (int j = 0; j < kTaps; ++j)
            for (int s = 0; s < 12; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int o = 32 * mt + (lane & 31), ci = s + 12 * (lane >> 5);
                    if (o < D::c0)
                        blob[L.a_in + ((mt * kTaps + j) * 12 + s) * 64 + lane] =
                            (float)((double)cin.w[(o * kF + ci) * kTaps + j] * scale(cin, o));
                }
    put_shift_tiles(cin, D::c0, D::mt0, blob + L.s_in);

    auto pack_dense = [&](int b) {
        for (int i = 0; i < D::nl; ++i) {
            ConvBn c = next();                       // weight (g, C_in, 3)
            const int cn = D::c0 + D::g * i;
            float* wl = blob + L.dense[b] + dense_layer_offset(D::c0, D::g, i);
            for (int ch = 0; ch < cn; ++ch)
                for (int o = 0; o < D::g; ++o)
                    for (int j = 0; j < 3; ++j)
                        wl[ch * 12 + o * 3 + j] = (float)((double)c.w[(o * cn + ch) * 3 + j] * scale(c, o));
            for (int o = 0; o < D::g; ++o) wl[cn * 12 + o] = (float)shift(c, o);
        }
    };
    auto pack_1x1 = [&](const ConvBn& c, int n_out, int n_tiles, float* a, float* s) {   // weight (n_out, c1, 1)
        for (int mt = 0; mt < n_tiles; ++mt)
            for (int ks = 0; ks < D::ks1; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int o = 32 * mt + (lane & 31), ch = 2 * ks + (lane >> 5);
                    if (o < n_out) a[(mt * D::ks1 + ks) * 64 + lane] = (float)((double)c.w[o * D::c1 + ch] * scale(c, o));
                }
        put_shift_tiles(c, n_out, n_tiles, s);
    };

    pack_dense(0);
    ConvBn tr = next();
    pack_1x1(tr, D::c0, D::mt0, blob + L.a_tr, blob + L.s_tr);
    pack_dense(1);
    ConvBn co = next();
    pack_1x1(co, D::c1, D::mt1, blob + L.a_out, blob + L.s_out);
    const float* wout = t[idx];                      // out.weight (2, c1)
    for (int mt = 0; mt < D::mt1; ++mt)
        for (int cl = 0; cl < 2; ++cl)
            for (int hh = 0; hh < 2; ++hh)
                for (int i = 0; i < 16; ++i) {
                    const int ch = 32 * mt + mfma32_row(i, hh);
                    if (ch < D::c1) blob[L.w_fin + ((mt * 2 + cl) * 2 + hh) * 16 + i] = wout[cl * D::c1 + ch];
                }
    return VAD_OK;
}

}  // namespace

// The stride-1 kernels use more than 64 KB of dynamic LDS: opt in once per device.
int ensure_densenet_attrs() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return VAD_OK;
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&densenet_stream_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DnStreamCfg<false>::lds_floats * (int)sizeof(float)));
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&densenet_stream_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DnStreamCfg<true>::lds_floats * (int)sizeof(float)));
    done[dev] = true;
    return VAD_OK;
}

namespace {

template <bool LARGE>
int launch(const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec, int mode,
           float threshold, hipStream_t st) {
    if (!wm.rows && wm.stride == kF) {
        using C = DnStreamCfg<LARGE>;
        int rc = ensure_densenet_attrs();
        if (rc != VAD_OK) return rc;
        const long blocks = (n_windows + C::windows - 1) / C::windows;
        VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
        hipLaunchKernelGGL(densenet_stream_kernel<LARGE>, dim3((unsigned)blocks), dim3(C::waves * 64),
                           (size_t)C::lds_floats * sizeof(float), st, x, n_windows, blob, p, dec, mode, threshold);
        VAD_HIP_CHECK(hipGetLastError());
        return VAD_OK;
    }
    const long blocks = (n_windows + 31) / 32;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
    hipLaunchKernelGGL(densenet_kernel<LARGE>, dim3((unsigned)blocks), dim3(64), 0, st, x, n_windows, wm, blob, p,
                       dec, mode, threshold);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

}  // namespace
}  // namespace vad

extern "C" size_t vad_densenet_blob_floats(int large) {
    return (size_t)(large ? vad::dn_layout<true>().total : vad::dn_layout<false>().total);
}

extern "C" int vad_densenet_tensor_count(int large) { return 5 * (3 + 2 * (large ? 8 : 6)) + 1; }

extern "C" int vad_pack_densenet_weights(int large, const float* const* tensors_host, int n_tensors, float* blob_host,
                                         size_t blob_floats) {
    using namespace vad;
    VAD_REQUIRE(tensors_host && blob_host, "null pointer argument");
    VAD_REQUIRE(n_tensors == vad_densenet_tensor_count(large), "expected %d tensors, got %d",
                vad_densenet_tensor_count(large), n_tensors);
    for (int i = 0; i < n_tensors; ++i) VAD_REQUIRE(tensors_host[i], "tensor %d is null", i);
    VAD_REQUIRE(blob_floats >= vad_densenet_blob_floats(large), "blob too small: need %zu floats",
                vad_densenet_blob_floats(large));
    return large ? pack<true>(tensors_host, blob_host) : pack<false>(tensors_host, blob_host);
}

namespace vad {
namespace {
int densenet_forward(const float* x_dev, int64_t n_windows, WindowMap wm, const float* blob_dev, int large, float* p_dev,
                     uint8_t* decision_dev, int decision_mode, float threshold, void* stream) {
    VAD_REQUIRE(x_dev && blob_dev && p_dev, "null pointer argument");
    VAD_REQUIRE(n_windows >= 0, "n_windows must be >= 0");
    VAD_REQUIRE(wm.stride >= 0 && (wm.stride % 4) == 0, "window_stride must be a non-negative multiple of 4 floats");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0, "x_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(p_dev) & 7) == 0, "p_dev must be 8-byte aligned");
    VAD_REQUIRE(decision_mode == VAD_DECIDE_ARGMAX || decision_mode == VAD_DECIDE_THRESHOLD, "unknown decision_mode %d",
                decision_mode);
    if (n_windows == 0) return VAD_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return large ? launch<true>(x_dev, n_windows, wm, blob_dev, p_dev, decision_dev, decision_mode, threshold, st)
                 : launch<false>(x_dev, n_windows, wm, blob_dev, p_dev, decision_dev, decision_mode, threshold, st);
}
}  // namespace
}  // namespace vad

extern "C" int vad_densenet_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                                        const float* blob_dev, int large, float* p_dev, uint8_t* decision_dev,
                                        int decision_mode, float threshold, void* stream) {
    return vad::densenet_forward(x_dev, n_windows, vad::WindowMap{(long)window_stride, nullptr, 0}, blob_dev, large, p_dev,
                                 decision_dev, decision_mode, threshold, stream);
}

extern "C" int vad_densenet_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                             int64_t n_windows, const float* blob_dev, int large, float* p_dev,
                                             uint8_t* decision_dev, int decision_mode, float threshold, void* stream) {
    vad::WindowMap wm;
    const int rc = vad::window_map_for_rows(n_rows, window_rows_dev, n_windows, &wm);
    if (rc != VAD_OK) return rc;
    return vad::densenet_forward(feat_dev, n_windows, wm, blob_dev, large, p_dev, decision_dev, decision_mode, threshold,
                                 stream);
}
