// Device-side building blocks of the conv + GRU window classifier (kernels: nicknet.hip, nicknet_stream.hip).
#pragma once
#include <math.h>
#include <string.h>

#include <type_traits>

#include "rnn_device.h"

namespace vad {
namespace nn {

using rnn::f16x8;
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
// A table of one product: [group of 4 k-steps][out block][64 lanes][4 k-steps] floats; a last group of one or two
// k-steps is stored at that width ([out block][64 lanes][1 or 2]), a last group of three at four.
__host__ __device__ constexpr int last_width(int ks) { return ks % 4 == 1 ? 1 : (ks % 4 == 2 ? 2 : 4); }
__host__ __device__ constexpr int table_floats(int ks, int nout) {
    return (groups(ks) - 1) * nout * 256 + nout * 64 * last_width(ks);
}
// float offset of the entry of (group g, out block o, lane) and its width
__host__ __device__ constexpr int entry_offset(int ks, int nout, int g, int o, int lane) {
    return g < groups(ks) - 1 ? ((g * nout + o) * 64 + lane) * 4
                              : (groups(ks) - 1) * nout * 256 + (o * 64 + lane) * last_width(ks);
}
constexpr int kBiasTile = 32;              // [2 blocks][4 q][4 i]
constexpr int kSplitBlock = 3 * 64 * 4;    // dwords of one out block of a float16 split table: 3 terms x 64 lanes x 16 bytes

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
    int gru_ih[2];       // out blocks = (r blocks, z blocks, n blocks); layers above the first: float16 split, as gru_hh
    int gru_hh[2];       // recurrent weights as float16 split A operands: [out block][hi, mid, lo][64 lanes] x 8 float16
    int conv_bias[4];    // [value, gate] bias tiles
    int gru_bias[2];     // [r, z, n-input, n-hidden] bias tiles
    int gru_scale[2];    // 2^-(14 + Sw), 2^Sw of the layer's recurrent table, then the same of its split input table
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
        at += g == 0 ? table_floats(tile_ksteps(D::gru_in(0)), gout) : gout * kSplitBlock;
        L.gru_hh[g] = at;
        at += gout * kSplitBlock;
    }
    for (int l = 0; l < D::nc; ++l) {
        L.conv_bias[l] = at;
        at += 2 * kBiasTile;
    }
    for (int g = 0; g < D::layers; ++g) {
        L.gru_bias[g] = at;
        at += 4 * kBiasTile;
    }
    for (int g = 0; g < D::layers; ++g) {
        L.gru_scale[g] = at;
        at += 4;
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
__device__ __forceinline__ void mm(f32x4 (&acc)[NACC], const float* tab, int lane, const float (&bv)[KS]) {
    constexpr int G = groups(KS), LW = last_width(KS);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        f32x4 av[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float* e = tab + entry_offset(KS, NOUT, g, o, 0);
            if (g < G - 1 || LW == 4) {
                av[o] = *reinterpret_cast<const f32x4*>(e + 4 * lane);
            } else if (LW == 2) {
                const float2 t = *reinterpret_cast<const float2*>(e + 2 * lane);
                av[o] = f32x4{t.x, t.y, 0.0f, 0.0f};
            } else {
                av[o] = f32x4{e[lane], 0.0f, 0.0f, 0.0f};
            }
        }
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
    const int lane = (int)(a_lane - reinterpret_cast<const f32x4*>(w));
    const float* a = w + LY.conv[L];
    if (has_m1) mm<NOUT, KS>(acc, a, lane, bm1);
    mm<NOUT, KS>(acc, a + LY.conv_tap[L], lane, b0);
    if (has_p1) mm<NOUT, KS>(acc, a + 2 * LY.conv_tap[L], lane, bp1);
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

// ---- products on the 16-bit matrix pipe ----------------------------------------------------------------
// A float32 product as six float16 partial products (rnn_device.h): a hidden state, times 2^14, and a weight table,
// times 2^Sw, are each the exact sum of three float16 terms; the products left out are below 2^-35 of a product and
// the accumulation is float32.  A lane's eight values of a tile are exactly its k-slots 8 q + j of
// v_mfma_f32_16x16x32_f16 (register [j >> 2][j & 3]), so a whole K = 32 is one instruction of 16 cycles where the
// float32 MFMA takes eight of 32 -- and this pipe runs beside the vector ALU.  Used where the B operand is a hidden
// state (|h| < 1): the GRU's recurrent products and the input products of the layers above the first.
struct SplitTile {
    rnn::u32x4 hi, mid, lo;          // dword jp = k-slots 2 jp (low half) and 2 jp + 1
};

__device__ __forceinline__ SplitTile split_state(const Tile& h) {
    SplitTile s;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const int j = 2 * jp;
        const rnn::SplitTerms t = rnn::split_pair<true, true>(h.b[j >> 2][j & 3] * rnn::kStateScale,
                                                              h.b[j >> 2][(j & 3) + 1] * rnn::kStateScale);
        s.hi[jp] = t.hi;
        s.mid[jp] = t.mid;
        s.lo[jp] = t.lo;
    }
    return s;
}

// acc[AT(o)] += unscale * (split table(out block o) . b); `tab`: [out block][hi, mid, lo][64 lanes] x 16 bytes
template <int NOUT, typename AT = Straight, int NACC>
__device__ __forceinline__ void mm_split(f32x4 (&acc)[NACC], const float* tab, int lane, const SplitTile& b, float unscale) {
    const rnn::u32x4* a16 = reinterpret_cast<const rnn::u32x4*>(tab) + lane;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const rnn::u32x4 ah = a16[(o * 3 + 0) * 64], am = a16[(o * 3 + 1) * 64], al = a16[(o * 3 + 2) * 64];
        f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
        c = MFMA_F16(ah, b.lo, c);                            // smallest terms first
        c = MFMA_F16(al, b.hi, c);
        c = MFMA_F16(am, b.mid, c);
        c = MFMA_F16(ah, b.mid, c);
        c = MFMA_F16(am, b.hi, c);
        c = MFMA_F16(ah, b.hi, c);
        f32x4& dst = acc[AT::at(o)];
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = fmaf(c[i], unscale, dst[i]);
    }
}

// One GRU cell step of one layer (torch.nn.GRU, gate order r, z, n): h <- (1 - z) n + z h.  `hs` is the split of h
// (of the previous state on entry, of the new one on return).  The first layer's input is the trunk's output `xin`
// (float32 MFMA); a layer above takes the split `xs` of the new state of the layer below.
template <bool LARGE, int LAYER>
__device__ __forceinline__ void gru_step(Tile& h, SplitTile& hs, const float (&xin)[tile_ksteps(Dims<LARGE>::gru_in(0))],
                                         const SplitTile& xs, const f32x4* a_lane, const float* w, bool first, int q) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    constexpr int NB = n_blocks(D::hidden);
    f32x4 acc[4 * NB];                                        // r, z, n-input, n-hidden
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[a * NB + blk] = bias_block(w + LY.gru_bias[LAYER] + a * kBiasTile, blk, q);
    const int lane = (int)(a_lane - reinterpret_cast<const f32x4*>(w));
    if constexpr (LAYER == 0) mm<3 * NB, tile_ksteps(D::gru_in(0))>(acc, w + LY.gru_ih[0], lane, xin);
    else mm_split<3 * NB>(acc, w + LY.gru_ih[LAYER], lane, xs, w[LY.gru_scale[LAYER] + 2]);
    if (!first) mm_split<3 * NB, SkipThird<NB>>(acc, w + LY.gru_hh[LAYER], lane, hs, w[LY.gru_scale[LAYER]]);    // h_0 = 0
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = sigmoid_pre(acc[blk][i]);
            const float z = sigmoid_pre(acc[NB + blk][i]);
            const float n = tanh_pre(fmaf(r, acc[3 * NB + blk][i], acc[2 * NB + blk][i]));
            h.b[blk][i] = fmaf(z, h.b[blk][i] - n, n);
        }
    hs = split_state(h);
}

// All GRU layers for one position: `x` is the trunk's output there.
template <bool LARGE>
__device__ __forceinline__ void gru_layers(Tile (&h)[Dims<LARGE>::layers], SplitTile (&hs)[Dims<LARGE>::layers], const Tile& x,
                                           const f32x4* a_lane, const float* w, bool first, int q) {
    using D = Dims<LARGE>;
    float xin[tile_ksteps(D::gru_in(0))];
    tile_kvals<D::gru_in(0)>(x, xin);
    gru_step<LARGE, 0>(h[0], hs[0], xin, hs[0], a_lane, w, first, q);
    if constexpr (D::layers > 1) {
        NN_STAGE_FENCE;
        gru_step<LARGE, 1>(h[1], hs[1], xin, hs[0], a_lane, w, first, q);
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
            mm<NB, KS>(acc, blob + LY.emb[0] + l * LY.emb_tab[0], lane, hv);
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
        mm<NB, KS>(acc, blob + LY.emb[1], lane, ev);
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
    mm<1, KS>(acc, blob + LY.emb[D::ne], lane, ev);
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
