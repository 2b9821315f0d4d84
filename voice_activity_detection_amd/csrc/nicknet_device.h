// Device-side building blocks of the conv + GRU window classifier (kernels: nicknet.hip, nicknet_stream.hip).
#pragma once
#include <math.h>
#include <string.h>

#include <type_traits>

#include "rnn_device.h"

namespace vad {
namespace nn {

using rnn::sigmoid_pre;
using rnn::tanh_pre;

constexpr int kT = VAD_FRAMES;
constexpr int kF = VAD_FEATURES;
constexpr int kWaveWindows = 16;
constexpr int kWaves = 8;                  // per block: two per SIMD (4 waves: +6 % time, 12: no change)
constexpr int kXk = kF / 4;                // k-steps of the feature rows: lane group q supplies feature 6 q + s

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- tiles ------------------------------------------------------------------------------------------
// nu units in blocks of 16; block b's unit 4 i + q is register [b][i] of the lanes of group q, so its first
// ceil(units / 4) k-steps carry all of them and the others are skipped.
__host__ __device__ constexpr int n_blocks(int nu) { return (nu + 15) / 16; }
__host__ __device__ constexpr int block_units(int nu, int b) { return nu - 16 * b > 16 ? 16 : (nu - 16 * b < 0 ? 0 : nu - 16 * b); }
__host__ __device__ constexpr int block_ksteps(int nu, int b) { return (block_units(nu, b) + 3) / 4; }
__host__ __device__ constexpr int tile_ksteps(int nu) { return block_ksteps(nu, 0) + block_ksteps(nu, 1); }
__host__ __device__ constexpr int groups(int ks) { return (ks + 3) / 4; }
// A table of one product: [group of 4 k-steps][out block][64 lanes][4 k-steps] floats
__host__ __device__ constexpr int table_floats(int ks, int nout) { return groups(ks) * nout * 256; }
constexpr int kBiasTile = 32;              // [2 blocks][4 q][4 i]

template <bool LARGE>
struct Dims {
    static constexpr int nc = LARGE ? 4 : 3;            // convolution layers        core/models.py:264-275
    static constexpr int layers = LARGE ? 2 : 1;        // GRU layers                core/models.py:94
    static constexpr int hidden = LARGE ? 29 : 21;      // last channels + 11 / + 5  core/models.py:303
    static constexpr int ne = LARGE ? 2 : 1;            // Linear + BN + ReLU stages of embed before its last Linear
    __host__ __device__ static constexpr int chan(int l) {          // channels after conv layer l; l = 0: the features
        return LARGE ? (l == 0 ? kF : l == 1 ? 32 : l == 2 ? 28 : l == 3 ? 25 : 18) : (l == 0 ? kF : l == 1 ? 20 : l == 2 ? 18 : 16);
    }
    __host__ __device__ static constexpr int conv_ks(int l) { return l == 0 ? kXk : tile_ksteps(chan(l)); }   // input of conv layer l (0-based)
    __host__ __device__ static constexpr int conv_nout(int l) { return 2 * n_blocks(chan(l + 1)); }          // value blocks, gate blocks
    __host__ __device__ static constexpr int gru_in(int layer) { return layer == 0 ? chan(nc) : hidden; }
    __host__ __device__ static constexpr int embed_in(int s) { return s == 0 ? hidden : (LARGE && s == 1 ? 32 : 18); }   // units of ONE input tile
    __host__ __device__ static constexpr int embed_out(int s) { return s == ne ? 2 : (LARGE && s == 0 ? 32 : 18); }
    static constexpr int head_cols = layers * hidden;
    static constexpr int n_tensors = 12 * nc + 4 * layers + 6 * ne + 2;
};

// Blob layout in floats.  [0, lds) is what a block copies into LDS.
struct NnLayout {
    int conv[4];         // three tap tables of conv layer l, conv_tap[l] floats each, out blocks = (value blocks, gate blocks)
    int conv_tap[4];
    int gru_ih[2];       // out blocks = (r blocks, z blocks, n blocks)
    int gru_hh[2];
    int conv_bias[4];    // [value, gate] bias tiles
    int gru_bias[2];     // [r, z, n-input, n-hidden] bias tiles
    int lds;
    int emb[3];          // embed stage s: one table per input tile (stage 0: one per GRU layer), emb_tab[s] floats each
    int emb_tab[3];
    int emb_bias[3];
    int total;
};

template <bool LARGE>
__host__ __device__ constexpr NnLayout nn_layout() {
    using D = Dims<LARGE>;
    NnLayout L{};
    int at = 0;
    for (int l = 0; l < D::nc; ++l) {
        L.conv[l] = at;
        L.conv_tap[l] = table_floats(D::conv_ks(l), D::conv_nout(l));
        at += 3 * L.conv_tap[l];
    }
    constexpr int gout = 3 * n_blocks(D::hidden);
    for (int g = 0; g < D::layers; ++g) {
        L.gru_ih[g] = at;
        at += table_floats(tile_ksteps(D::gru_in(g)), gout);
        L.gru_hh[g] = at;
        at += table_floats(tile_ksteps(D::hidden), gout);
    }
    for (int l = 0; l < D::nc; ++l) {
        L.conv_bias[l] = at;
        at += 2 * kBiasTile;
    }
    for (int g = 0; g < D::layers; ++g) {
        L.gru_bias[g] = at;
        at += 4 * kBiasTile;
    }
    L.lds = at;
    for (int s = 0; s <= D::ne; ++s) {
        L.emb[s] = at;
        L.emb_tab[s] = table_floats(tile_ksteps(D::embed_in(s)), n_blocks(D::embed_out(s)));
        at += (s == 0 ? D::layers : 1) * L.emb_tab[s];
        L.emb_bias[s] = at;
        at += kBiasTile;
    }
    L.total = at;
    return L;
}

struct Tile {
    f32x4 b[2];
};

// Between the pipeline's stages: left free, the scheduler starts the next stage's operand reads (24 registers per
// group of k-steps) under the current one, which costs the large model registers it does not have.
#define NN_STAGE_FENCE __builtin_amdgcn_sched_barrier(0)

template <int NU>
__device__ __forceinline__ void tile_kvals(const Tile& t, float (&v)[tile_ksteps(NU)]) {
    constexpr int k0 = block_ksteps(NU, 0), k1 = block_ksteps(NU, 1);
#pragma unroll
    for (int i = 0; i < k0; ++i) v[i] = t.b[0][i];
#pragma unroll
    for (int i = 0; i < k1; ++i) v[k0 + i] = t.b[1][i];
}

// acc[AT(o)] += table(out block o) . in, over KS k-steps; `a` is the table's first entry of this lane
struct Straight {
    static __device__ __forceinline__ constexpr int at(int o) { return o; }
};
template <int NB>
struct SkipThird {      // r, z, then the n gate's hidden part: output blocks 2 NB .. of the table land NB further on
    static __device__ __forceinline__ constexpr int at(int o) { return o < 2 * NB ? o : o + NB; }
};

template <int NOUT, int KS, typename AT = Straight, int NACC>
__device__ __forceinline__ void mm(f32x4 (&acc)[NACC], const f32x4* a, const float (&bv)[KS]) {
#pragma unroll
    for (int g = 0; g < groups(KS); ++g) {
        f32x4 av[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) av[o] = a[(g * NOUT + o) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * g + j < KS) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) acc[AT::at(o)] = MFMA16(av[o][j], bv[4 * g + j], acc[AT::at(o)]);
            }
        }
    }
}

__device__ __forceinline__ f32x4 bias_block(const float* tile, int blk, int q) {
    return *reinterpret_cast<const f32x4*>(tile + (blk * 4 + q) * 4);
}

// One gated convolution at one position: tanh(BN(conv)) * sigmoid(BN(conv_gate)) of the three input positions
// (a tap whose position lies outside the window is zero padding and is skipped; wave-uniform).
template <bool LARGE, int L>
__device__ __forceinline__ Tile conv_layer(const f32x4* a_lane, const float* w, const float (&bm1)[Dims<LARGE>::conv_ks(L)],
                                           const float (&b0)[Dims<LARGE>::conv_ks(L)],
                                           const float (&bp1)[Dims<LARGE>::conv_ks(L)], bool has_m1, bool has_p1, int q) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    constexpr int NB = n_blocks(D::chan(L + 1)), NOUT = 2 * NB, KS = D::conv_ks(L);
    f32x4 acc[NOUT];
#pragma unroll
    for (int vg = 0; vg < 2; ++vg)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[vg * NB + blk] = bias_block(w + LY.conv_bias[L] + vg * kBiasTile, blk, q);
    const f32x4* a = a_lane + LY.conv[L] / 4;
    if (has_m1) mm<NOUT, KS>(acc, a, bm1);
    mm<NOUT, KS>(acc, a + LY.conv_tap[L] / 4, b0);
    if (has_p1) mm<NOUT, KS>(acc, a + 2 * (LY.conv_tap[L] / 4), bp1);
    Tile out;
    out.b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i) out.b[blk][i] = tanh_pre(acc[blk][i]) * sigmoid_pre(acc[NB + blk][i]);
    return out;
}

// Pipeline stage of conv layer LI >= 1 at iteration p: `cur` (layer LI - 1 at position p - LI + 1, or anything once
// that position is past the window) enters the ring, and layer LI's output at position p - LI replaces it.
template <bool LARGE, int LI>
__device__ __forceinline__ void conv_stage(Tile (&ring)[3], Tile& cur, int p, const f32x4* a_lane, const float* w, int q) {
    using D = Dims<LARGE>;
    ring[0] = ring[1];
    ring[1] = ring[2];
    ring[2] = cur;
    const int pos = p - LI;
    if (pos >= 0 && pos < kT) {
        float km[D::conv_ks(LI)], k0[D::conv_ks(LI)], kp[D::conv_ks(LI)];
        tile_kvals<D::chan(LI)>(ring[0], km);
        tile_kvals<D::chan(LI)>(ring[1], k0);
        tile_kvals<D::chan(LI)>(ring[2], kp);
        cur = conv_layer<LARGE, LI>(a_lane, w, km, k0, kp, pos >= 1, pos + 1 < kT, q);
    }
}

// One GRU cell step of one layer (torch.nn.GRU, gate order r, z, n): h <- (1 - z) n + z h.
template <bool LARGE, int LAYER>
__device__ __forceinline__ void gru_step(Tile& h, const float (&xin)[tile_ksteps(Dims<LARGE>::gru_in(LAYER))],
                                         const f32x4* a_lane, const float* w, bool first, int q) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    constexpr int NB = n_blocks(D::hidden), KSI = tile_ksteps(D::gru_in(LAYER)), KSH = tile_ksteps(D::hidden);
    f32x4 acc[4 * NB];                                        // r, z, n-input, n-hidden
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[a * NB + blk] = bias_block(w + LY.gru_bias[LAYER] + a * kBiasTile, blk, q);
    mm<3 * NB, KSI>(acc, a_lane + LY.gru_ih[LAYER] / 4, xin);
    if (!first) {                                             // h_0 = 0
        float hv[KSH];
        tile_kvals<D::hidden>(h, hv);
        mm<3 * NB, KSH, SkipThird<NB>>(acc, a_lane + LY.gru_hh[LAYER] / 4, hv);
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = sigmoid_pre(acc[blk][i]);
            const float z = sigmoid_pre(acc[NB + blk][i]);
            const float n = tanh_pre(fmaf(r, acc[3 * NB + blk][i], acc[2 * NB + blk][i]));
            h.b[blk][i] = fmaf(z, h.b[blk][i] - n, n);
        }
}

// embed (core/models.py:109-128, eval mode) on the final hidden states, softmax (core/models.py:330), decision.
// Its tables are read from the blob in global memory: once per 16 windows.
template <bool LARGE>
__device__ __forceinline__ void embed_head(const Tile (&h)[Dims<LARGE>::layers], const float* __restrict__ blob, int lane,
                                           long w, long n_windows, float* __restrict__ p_out, uint8_t* __restrict__ dec_out,
                                           int mode, float threshold, float* __restrict__ logits_out) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    const int q = lane >> 4, n = lane & 15;
    const f32x4* a_lane = reinterpret_cast<const f32x4*>(blob) + lane;
    Tile e;
    {   // stage 0: columns l * H + u of the head input are unit u of GRU layer l's final state
        constexpr int NB = n_blocks(D::embed_out(0)), KS = tile_ksteps(D::hidden);
        f32x4 acc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[blk] = bias_block(blob + LY.emb_bias[0], blk, q);
#pragma unroll
        for (int l = 0; l < D::layers; ++l) {
            float hv[KS];
            tile_kvals<D::hidden>(h[l], hv);
            mm<NB, KS>(acc, a_lane + (LY.emb[0] + l * LY.emb_tab[0]) / 4, hv);
        }
        e.b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) e.b[blk][i] = fmaxf(acc[blk][i], 0.0f);
    }
    if (LARGE) {
        constexpr int NB = n_blocks(D::embed_out(1)), KS = tile_ksteps(D::embed_in(1));
        f32x4 acc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[blk] = bias_block(blob + LY.emb_bias[1], blk, q);
        float ev[KS];
        tile_kvals<D::embed_in(1)>(e, ev);
        mm<NB, KS>(acc, a_lane + LY.emb[1] / 4, ev);
        e.b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) e.b[blk][i] = fmaxf(acc[blk][i], 0.0f);
    }
    constexpr int KS = tile_ksteps(D::embed_in(D::ne));
    f32x4 acc[1];
    acc[0] = bias_block(blob + LY.emb_bias[D::ne], 0, q);
    float ev[KS];
    tile_kvals<D::embed_in(D::ne)>(e, ev);
    mm<1, KS>(acc, a_lane + LY.emb[D::ne] / 4, ev);
    // logit 0 is unit 0 = register 0 of lane group 0, logit 1 is unit 1 = register 0 of lane group 1
    const float l0 = __shfl(acc[0][0], n, 64), l1 = __shfl(acc[0][0], 16 + n, 64);
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    if (q == 0 && w < n_windows) {
        if (logits_out) *reinterpret_cast<float2*>(logits_out + 2 * w) = make_float2(l0, l1);
        if (p_out) *reinterpret_cast<float2*>(p_out + 2 * w) = make_float2(p0, p1);
        if (dec_out) dec_out[w] = (mode == VAD_DECIDE_THRESHOLD) ? (p1 >= threshold) : (p1 > p0);
    }
}

__device__ __forceinline__ void load_x_row(float (&v)[kXk], const float* p) {      // 6 floats, 8-byte aligned
#pragma unroll
    for (int j = 0; j < kXk / 2; ++j) {
        const float2 t = *reinterpret_cast<const float2*>(p + 2 * j);
        v[2 * j] = t.x;
        v[2 * j + 1] = t.y;
    }
}


// The stride-1 kernel (nicknet_stream.hip): windows w = 0 .. n_windows - 1 start at rows w of a packed (F, 24) matrix.
int launch_stream(bool large, const float* feat, long n_windows, const float* blob, float* p, uint8_t* dec, int mode,
                  float threshold, int blocks, hipStream_t st);
int stream_attrs();    // opts the stride-1 kernels in to their LDS size (current device)

}  // namespace nn
}  // namespace vad
