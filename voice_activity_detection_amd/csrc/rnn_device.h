// Device-side building blocks of the GRU / LSTM window models (kernels and launch code: rnn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "vad_internal.h"

namespace vad {
namespace rnn {

constexpr int kT = VAD_FRAMES;      // 30 steps
constexpr int kF = VAD_FEATURES;    // 24 features
constexpr int kKs = kF / 2;         // 12 k-steps of the input projection
constexpr int kHp = 32;             // padded hidden size (one 32-row tile per gate)
constexpr int kHk = kHp / 2;        // 16 k-steps of the recurrent product
constexpr int kL1 = 26;             // lin1 outputs
constexpr int kWindowsPerWave = 32;
constexpr int kWavesPerBlock = 4;   // rnn_kernel

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Blob layout in floats; shared by the host packer and the kernels.  `ut` = number of 32-unit
// tiles covering the hidden size (1 for the reference's H = 30, 3 for H = 90).
struct NetLayout {
    int gates;        // 3 GRU, 4 LSTM
    int ut;
    int a_ih;         // [gates][ut][12][64]        A operands of the input projection
    int a_hh;         // [gates][ut][ut*16][64]     A operands of the recurrent product
    int cb;           // [4][ut][2][16]             accumulator initial values (biases), per lane half
    int head;         // a1 [30][ut*16][64]         A operands of lin1 (large) or lin (small, rows 0-1)
    int tail;         // large: b1e [2][16], w2e [2][2][16], b2 [2], 2 pad; small: bl [2], 2 pad; then for the float16 tables
                      // 2^S, 2^-S, 2^Sw, pad (kTailScale: S = kStateShift + Sw, the accumulators' scale; see split_f16)
    int s_hh;         // ut == 1: [gates][2][3][64] x 8 bf16   a_hh split into bf16 (hi, mid, lo) A operands;
                      // ut > 1 (GRU): wide_split_gate_entries(ut) entries of [64] x 8 bf16, order above
    int s_head;       // ut == 1: [30][2][3][64] x 8 bf16      the head's A operands, split the same way;
                      // ut > 1: [30][2 mi][ut kb][3 terms][64] x 8 bf16
    int h_hh;         // the same two tables with the operands split into three float16 (split_f16 below)
    int h_head;
    int total;
};

__host__ __device__ constexpr int kTailScale(bool large) { return large ? 2 * 16 + 2 * 2 * 16 + 4 : 4; }   // floats of the tail before the scales
constexpr int kStateShift = 14;              // float16 modes keep the hidden state as 2^14 h
constexpr float kStateScale = 16384.0f;      // 2^kStateShift
constexpr float kStateUnscale = 1.0f / 16384.0f;
constexpr int kSplitTile = 2 * 3 * 64 * 4;   // dwords of one split 32x32 A tile: 2 halves of 16 rows, 3 terms, 16 B per lane
constexpr int kSplitEntry = 64 * 4;          // dwords of one bf16 A operand of v_mfma_f32_16x16x32_bf16 (16 rows x 32 k)

// GRU with ut > 1 unit tiles: the split recurrent operands are entries (gate g, unit tile mt, row half mi, k-block kb,
// term) in two sections, first hi and mid of everything, then the lo terms gate by gate (r, z, n): the wide split
// kernel keeps the leading wide_split_lds_entries of them in LDS.
__host__ __device__ constexpr int wide_split_hm_entries(int ut) { return 3 * ut * 2 * ut * 2; }
__host__ __device__ constexpr int wide_split_gate_entries(int ut) { return 3 * ut * 2 * ut * 3; }
// 158 entries of 1 KB and the biases are what 160 KB of LDS take: for three unit tiles all but the last four lo
// entries of the last gate (read from the weight blob instead)
__host__ __device__ constexpr int wide_split_lds_entries(int ut) {
    return wide_split_gate_entries(ut) < 158 ? wide_split_gate_entries(ut) : 158;
}
__host__ __device__ constexpr int wide_split_base(int ut, int g, int mt, int mi, int kb) {
    return (((g * ut + mt) * 2 + mi) * ut + kb);
}

__host__ __device__ inline NetLayout net_layout(bool lstm, bool large, int ut = 1) {
    NetLayout L;
    L.gates = lstm ? 4 : 3;
    L.ut = ut;
    L.a_ih = 0;
    L.a_hh = L.a_ih + L.gates * ut * kKs * 64;
    L.cb = L.a_hh + L.gates * ut * ut * kHk * 64;
    L.head = L.cb + 4 * ut * 2 * 16;
    L.tail = L.head + kT * ut * kHk * 64;
    L.s_hh = L.tail + kTailScale(large) + 4;
    const int hh_floats = ut == 1 ? L.gates * kSplitTile : wide_split_gate_entries(ut) * kSplitEntry;
    const int head_floats = ut == 1 ? kT * kSplitTile : kT * 2 * ut * 3 * kSplitEntry;
    L.s_head = L.s_hh + hh_floats;
    L.h_hh = L.s_head + head_floats;
    L.h_head = L.h_hh + hh_floats;
    L.total = L.h_head + head_floats;
    return L;
}

__host__ __device__ inline int unit_tiles(int hidden) { return (hidden + 31) / 32; }

// Placement of hidden units on the 32 rows of a tile.  K-step i of the next step's product takes rows
// (i & 3) + 8 (i >> 2) and that + 4 as its two k values.  A tile that holds nu < 32 units puts them, in order,
// on the rows of its first ceil(nu / 2) k-steps, so that the remaining k-steps multiply padding only and the
// kernels skip them: nu = 30 (the reference's hidden size) leaves out rows 27 and 31 = k-step 15; the last
// tile of hidden = 90 holds 26 units on rows 0..24 and 28 = k-steps 0..12.
__host__ __device__ inline int tile_units(int hidden, int tile) {
    const int n = hidden - 32 * tile;
    return n > 32 ? 32 : (n < 0 ? 0 : n);
}
__host__ __device__ inline int tile_k_steps(int nu) { return (nu + 1) / 2; }
__host__ __device__ inline int k_step_of_row(int r) { return (r & 3) + 4 * (r >> 3); }
__host__ __device__ inline int unit_of_tile_row(int r, int nu) {     // unit index inside the tile, or -1 = padding
    const int ks = tile_k_steps(nu);
    if (k_step_of_row(r) >= ks) return -1;
    int rank = 0;
    for (int q = 0; q < r; ++q) rank += k_step_of_row(q) < ks ? 1 : 0;
    return rank < nu ? rank : -1;
}
__host__ __device__ inline int hidden_k_steps(int hidden) { return tile_k_steps(tile_units(hidden, 0)); }   // one-tile models
__host__ __device__ inline int unit_of_row(int r, int hidden) { return unit_of_tile_row(r, tile_units(hidden, 0)); }

// The packer scales the sigmoid gates' weights and biases by -log2(e) and the tanh gates' by
// 2 log2(e), so the pre-activations arrive ready for v_exp_f32 (2^x).
constexpr double kSigmoidScale = -1.4426950408889634;
constexpr double kTanhScale = 2.8853900817779268;
__device__ __forceinline__ float sigmoid_pre(float a) {       // a = -x log2 e   ->  1 / (1 + e^-x)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
__device__ __forceinline__ float tanh_pre(float a) {          // a = 2 x log2 e  ->  1 - 2 / (1 + e^2x)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)), 1.0f);
}

// One LSTM unit from its pre-scaled pre-activations (a_i, a_f, a_o times -log2 e, a_g times 2 log2 e): returns
// `scale` * h' and sets c'.  With E_x = 2^a_x the cell is
//     i g  = (E_g - 1) / ((1 + E_i)(1 + E_g)),   c' = c / (1 + E_f) + i g,   h' = (E_c - 1) / ((1 + E_o)(1 + E_c)),  E_c = e^(2 c'):
// a product of a sigmoid and a tanh shares ONE reciprocal, so the unit costs 5 v_exp + 3 v_rcp instead of 5 + 5 (a
// transcendental takes the issue slot of four plain instructions) at one more plain instruction.  Overflows stay
// harmless: E_g is capped (its own infinity would meet a zero), an infinite E_i or E_o makes the reciprocal, and
// with it the product, zero as the closed gate demands, and E_c is finite because |c'| <= 30 steps.
static_assert(kT <= 40, "e^(2 c) must stay finite: |c| grows by at most 1 per step");
__device__ __forceinline__ float lstm_cell(float ai, float af, float ag, float ao, float c, float scale, float& c_new) {
    const float ei = __builtin_amdgcn_exp2f(ai);
    const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(af));
    const float eg = fminf(__builtin_amdgcn_exp2f(ag), 0x1p60f);
    const float eo = __builtin_amdgcn_exp2f(ao);
    const float tg = 1.0f + eg;
    const float igg = (eg - 1.0f) * __builtin_amdgcn_rcpf(fmaf(ei, tg, tg));
    const float cn = fmaf(fg, c, igg);
    c_new = cn;
    const float ec = __builtin_amdgcn_exp2f(2.8853900817779268f * cn);
    const float tc = 1.0f + ec;
    return fmaf(scale, ec, -scale) * __builtin_amdgcn_rcpf(fmaf(eo, tc, tc));
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---- fp32 products on the bf16 matrix pipe ---------------------------------------------------------
// A float32 is the exact sum of three bfloat16 (8 + 8 + 8 significant bits, by truncation: hi = the upper
// 16 bits, mid = the upper 16 bits of x - hi, lo = x - hi - mid).  x * y is then the exact sum of the nine
// bf16 x bf16 partial products, each of which the bf16 MFMA forms exactly and adds in float32.
// With all 9 terms the only rounding left is the float32 accumulation, as in a float32 FMA chain; with 8
// terms lo*lo is dropped (< 2^-30 of the product, 1/64 of the rounding of one float32 operation); with 6
// terms the three smallest (mid*lo, lo*mid, lo*lo, <= 2^-23 of the product) are dropped.  The bf16 matrix
// pipe does K = 32 of a 32 x 32 tile in 64 cycles against 16 x 64 cycles on the float32 MFMA, and unlike the
// float32 MFMA it runs beside vector instructions.
//
// The instruction is v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it than on the 32x32x16
// shape): a wave's 32 units x 32 windows are four 16 x 16 sub-tiles (mi, ni) = (unit half, window half), and the
// whole K = 32 of the hidden state is one instruction.  Lane l = 16 q + n holds, of sub-tile (mi, ni), tile rows
// 16 mi + 4 q + i (i = 0..3) of window 16 ni + n in accumulator element i: element (mi, ni, i) of a tile is
// index (2 mi + ni) * 4 + i of an f32x16.  As B operand the same lane supplies k-slots 8 q + j (j = 0..7) of
// window 16 ni + n; with k-slot 8 q + j := tile row 16 (j >> 2) + 4 q + (j & 3) those are exactly the eight rows
// the lane holds for that window, so the hidden state never leaves its lane, and the A operands carry their
// columns in that k-slot order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

#define MFMA_BF16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// ---- the same on float16 operands ---------------------------------------------------------------------
// float16 carries 11 significant bits against bfloat16's 8, so three terms overshoot float32's 24 bits and six
// partial products reach what nine bfloat16 ones do: with x = hi + mid + lo split by round-to-nearest
// (hi = f16(x), mid = f16(x - hi), lo = f16(x - hi - mid); |mid| <= 2^-12 |x|, |lo| <= 2^-24 |x|) the products left
// out of the six -- mid * lo, lo * mid, lo * lo -- are below 2^-35 of x * y.  float16's narrow exponent is dealt with
// by powers of two, which cost no rounding: three terms are exact while the last bit of x is at least 2^-24
// (float16's smallest subnormal), i.e. for 0.5 <= |x| <= 65504, so the hidden state is carried as H = 2^14 h
// (|h| <= 1: exact from |h| = 2^-15 up, and rounded to a 2^-39 grid below) and the packer multiplies the weights
// by the largest 2^Sw that keeps them under 65504.  The accumulators then hold 2^S times the pre-activations,
// S = 14 + Sw: the projected rows and biases they start from are multiplied by 2^S, and the gate math multiplies
// by 2^-S before its exponentials -- two vector multiplies per GRU unit and step, four per LSTM unit; everywhere
// else the power of two rides in a constant of an existing fused multiply-add.  'f16x4' keeps hi * hi, hi * mid,
// mid * hi, mid * mid of the same split: products to 2^-23, four instructions.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

template <bool F16>
__device__ __forceinline__ f32x4s mfma_split(const u32x4& a, const u32x4& b, const f32x4s& c) {
    if (F16) return MFMA_F16(a, b, c);
    return MFMA_BF16(a, b, c);
}

// (x0, x1) -> one dword per term, x0 in the low half.  `lo` is only formed when WITH_LO.
struct SplitTerms {
    unsigned hi, mid, lo;
};

template <bool F16, bool WITH_LO>
__device__ __forceinline__ SplitTerms split_pair(float x0, float x1) {
    unsigned hi, mid, lo = 0;
    if (F16) {
        // x - (float)f16(x) is exact; the residuals by v_fma_mix_f32, a float32 fma whose first operand is read as the
        // low / high float16 of a dword
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{x0, x1}, f16x2));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
        mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{r0, r1}, f16x2));
        if (WITH_LO) {
            float q0, q1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(mid), "v"(r0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(mid), "v"(r1));
            lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{q0, q1}, f16x2));
        }
    } else {
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u);
        const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
        // upper halves of (second, first) -> one dword, first in the low half
        hi = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        mid = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        if (WITH_LO) {
            const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u);
            const float q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
            lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
        }
    }
    return SplitTerms{hi, mid, lo};
}

// Host twin of split_pair: x -> the three 16-bit terms (hi, mid, lo), bfloat16 by truncation or float16 by
// round-to-nearest; *finite is cleared when a float16 term overflows.
inline void split3_host(float x, bool f16, uint16_t out[3], bool* finite) {
    float r = x;
    for (int term = 0; term < 3; ++term) {
        if (f16) {
            const _Float16 h = (_Float16)r;
            memcpy(&out[term], &h, 2);
            const float back = (float)h;
            if (!(back - back == 0.0f)) *finite = false;
            r -= back;
        } else {
            uint32_t bits;
            memcpy(&bits, &r, 4);
            const uint32_t hb = bits & 0xFFFF0000u;
            float hi;
            memcpy(&hi, &hb, 4);
            out[term] = (uint16_t)(hb >> 16);
            r -= hi;
        }
    }
}

__host__ __device__ inline int split_index(int mi, int ni, int i) { return (2 * mi + ni) * 4 + i; }
__host__ __device__ inline int split_row_of_k_slot(int q, int j) { return 16 * (j >> 2) + 4 * q + (j & 3); }

// B operands (hi, mid, lo) of the lane's two windows: [ni], dword jp = k-slots 2 jp (low half) and 2 jp + 1
struct SplitB {
    u32x4 hi[2], mid[2], lo[2];
};

template <int NI = 2, bool F16 = false, bool WITH_LO = true>
__device__ __forceinline__ void split_state(SplitB& b, const f32x16& h) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int j = 2 * jp;                                       // j and j + 1 share j >> 2
            const SplitTerms s = split_pair<F16, WITH_LO>(h[split_index(j >> 2, ni, j & 3)], h[split_index(j >> 2, ni, (j & 3) + 1)]);
            b.hi[ni][jp] = s.hi;
            b.mid[ni][jp] = s.mid;
            b.lo[ni][jp] = s.lo;
        }
}

// acc += A . B for one split 32 x 32 tile; `a` points at the tile's [2 mi][3 terms][64 lanes] 16-byte entries
// (LDS or global memory)
template <int TERMS, int NI = 2, bool F16 = false>
__device__ __forceinline__ f32x16 split_tile_mfma(f32x16 acc, const u32x4* a, const SplitB& b, int lane) {
    static_assert(TERMS == 4 || TERMS == 6 || TERMS == 8 || TERMS == 9, "4, 6, 8 or 9 partial products");
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const u32x4 ah = a[(mi * 3 + 0) * 64 + lane], am = a[(mi * 3 + 1) * 64 + lane];
        u32x4 al = ah;
        if (TERMS > 4) al = a[(mi * 3 + 2) * 64 + lane];
        f32x4s c[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[ni][i] = acc[split_index(mi, ni, i)];
        auto term = [&](const u32x4& at, const u32x4 (&bt)[2]) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) c[ni] = mfma_split<F16>(at, bt[ni], c[ni]);
        };
        if (TERMS == 9) term(al, b.lo);                     // smallest terms first
        if (TERMS >= 8) {
            term(am, b.lo);
            term(al, b.mid);
        }
        if (TERMS >= 6) {
            term(ah, b.lo);
            term(al, b.hi);
        }
        term(am, b.mid);
        term(ah, b.mid);
        term(am, b.hi);
        term(ah, b.hi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[split_index(mi, ni, i)] = c[ni][i];
    }
    return acc;
}

__device__ __forceinline__ f32x16 load16(const float* p) {      // 16 consecutive floats, 16-byte aligned
    const float4* s = reinterpret_cast<const float4*>(p);
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = s[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
    return r;
}

// Recurrent product for one step: acc += W_hh . h  (GRU keeps W_hn h + b_hn apart in acc[3]).
template <bool LSTM, int KS>
__device__ __forceinline__ void recurrent_mfma(f32x16 (&acc)[4], const f32x16& h, const float* a_hh, int lane) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const float hv = h[i];
        if (LSTM) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = MFMA(a_hh[(g * kHk + i) * 64 + lane], hv, acc[g]);
        } else {
            acc[0] = MFMA(a_hh[(0 * kHk + i) * 64 + lane], hv, acc[0]);
            acc[1] = MFMA(a_hh[(1 * kHk + i) * 64 + lane], hv, acc[1]);
            acc[3] = MFMA(a_hh[(2 * kHk + i) * 64 + lane], hv, acc[3]);
        }
    }
}

// Gate non-linearities, elementwise in (unit, window).
// (NI = 1: the split layout with one window half per lane uses elements (2 mi) * 4 + i only)
template <bool LSTM, int NI = 2>
__device__ __forceinline__ void gate_update(const f32x16 (&acc)[4], f32x16& h, f32x16& c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (NI == 1 && ((i >> 2) & 1)) continue;
        if (LSTM) {
            float cn;
            h[i] = lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[3][i], c[i], 1.0f, cn);
            c[i] = cn;
        } else {
            const float r = sigmoid_pre(acc[0][i]);
            const float z = sigmoid_pre(acc[1][i]);
            const float n = tanh_pre(fmaf(r, acc[3][i], acc[2][i]));
            h[i] = fmaf(z, h[i] - n, n);
        }
    }
}

// The same for the float16 modes: the accumulators hold 2^S times the pre-activations (`up` = 2^S, `inv` = 2^-S) --
// except the GRU's n-input part acc[2], which no matrix product touches -- and the state is H = 2^14 h.
// a[i] * s for the 16 elements of a tile
__device__ __forceinline__ f32x16 scale16(const f32x16& a, float s) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = a[i] * s;
    return r;
}

template <bool LSTM, int NI = 2>
__device__ __forceinline__ void gate_update_scaled(const f32x16 (&acc_in)[4], f32x16& h, f32x16& c, float up, float inv) {
    // back to the pre-activations' own scale, except the GRU's hidden n part, whose 2^-S rides on r
    f32x16 acc[4];
    acc[0] = scale16(acc_in[0], inv);
    acc[1] = scale16(acc_in[1], inv);
    acc[2] = LSTM ? scale16(acc_in[2], inv) : acc_in[2];
    acc[3] = LSTM ? scale16(acc_in[3], inv) : acc_in[3];
    inv = 1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (NI == 1 && ((i >> 2) & 1)) continue;
        if (LSTM) {
            float cn;
            h[i] = lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[3][i], c[i], kStateScale, cn);      // 2^14 h
            c[i] = cn;
        } else {
            const float r_inv = __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(acc[0][i] * inv), up, up));   // 2^-S r
            const float z = sigmoid_pre(acc[1][i] * inv);
            const float n = fmaf(-2.0f * kStateScale,
                                 __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r_inv, acc[3][i], acc[2][i]))), kStateScale);
            h[i] = fmaf(z, h[i] - n, n);
        }
    }
}

// The step-t slice of the flattened head (lin1: 26 rows, or lin: 2 rows): out[o] += W[o][t*H + u] h_t[u].
// Its weights sit in the blob as MFMA A operands, a1[(t*16 + i)*64 + lane] = W[lane & 31][t*H + u(i, lane >> 5)].
// The 26-row head is one more matrix-core tile.  The 2-row head would fill 2 of a tile's 32 rows, so it runs
// on the vector ALU instead: each lane multiplies its own 16 units into two running sums (the weights of
// rows 0 and 1 are the adjacent floats at lane positions 32*half and 32*half + 1, the same for the whole
// half-wave) and the two half-waves are added once, after the last step (head_finish).
template <bool LARGE, int KS>
struct HeadWeights {
    float a[LARGE ? KS : 1];
    float2 w[LARGE ? 1 : KS];
};

template <bool LARGE, int KS>
__device__ __forceinline__ void head_load(HeadWeights<LARGE, KS>& hw, const float* __restrict__ a1, int t, int lane) {
    if (LARGE) {
#pragma unroll
        for (int i = 0; i < KS; ++i) hw.a[i] = a1[(t * kHk + i) * 64 + lane];
    } else {
        const float2* p = reinterpret_cast<const float2*>(a1 + (long)t * kHk * 64 + 32 * (lane >> 5));
#pragma unroll
        for (int i = 0; i < KS; ++i) hw.w[i] = p[i * 32];
    }
}

template <bool LARGE, int KS>
__device__ __forceinline__ void head_step(f32x16& lacc, const HeadWeights<LARGE, KS>& hw, const f32x16& h) {
    if (LARGE) {
#pragma unroll
        for (int i = 0; i < KS; ++i) lacc = MFMA(hw.a[i], h[i], lacc);
    } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            lacc[0] = fmaf(hw.w[i].x, h[i], lacc[0]);
            lacc[1] = fmaf(hw.w[i].y, h[i], lacc[1]);
        }
    }
}

// after the last step: lacc in the accumulator layout head_epilogue reads
template <bool LARGE>
__device__ __forceinline__ void head_finish(f32x16& lacc) {
    if (!LARGE) {
        lacc[0] += __shfl_xor(lacc[0], 32, 64);
        lacc[1] += __shfl_xor(lacc[1], 32, 64);
    }
}

// (ReLU, lin2) or the bias of lin; softmax over the 2 classes; decision.
template <bool LARGE>
__device__ __forceinline__ void head_epilogue(const f32x16& lacc, const float* __restrict__ tail, int half, long w,
                                              long n_windows, float* __restrict__ p_out, uint8_t* __restrict__ dec_out,
                                              int mode, float threshold) {
    float l0, l1;
    if (LARGE) {
        const float* b1e = tail + half * 16;
        const float* w2a = tail + 32 + (0 * 2 + half) * 16;
        const float* w2b = tail + 32 + (1 * 2 + half) * 16;
        l0 = l1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = fmaxf(lacc[i] + b1e[i], 0.0f);
            l0 = fmaf(v, w2a[i], l0);
            l1 = fmaf(v, w2b[i], l1);
        }
        l0 += __shfl_xor(l0, 32, 64);
        l1 += __shfl_xor(l1, 32, 64);
        l0 += tail[96];
        l1 += tail[97];
    } else {
        l0 = lacc[0] + tail[0];          // rows 0 and 1 of the tile live in lane half 0
        l1 = lacc[1] + tail[1];
    }
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    if (half == 0 && w < n_windows) {
        *reinterpret_cast<float2*>(p_out + 2 * w) = make_float2(p0, p1);
        if (dec_out) dec_out[w] = (mode == VAD_DECIDE_THRESHOLD) ? (p1 >= threshold) : (p1 > p0);
    }
}

// ---- the head in the split (16 x 16 sub-tile) layout ------------------------------------------------
// Tile row p = 16 mi + 4 q + i of the float32 images (biases, lin1 / lin2 rows, the 2-row head's weights) sits at
// element (p & 3) + 4 (p >> 3) of lane half (p >> 2) & 1: for a lane's four rows i = 0..3 that is half q & 1,
// elements 4 (2 mi + (q >> 1)) + i, a contiguous float4.
__device__ __forceinline__ int split_image_half(int q) { return q & 1; }
__device__ __forceinline__ int split_image_elem(int q, int mi) { return 4 * (2 * mi + (q >> 1)); }

// 2-row head on the vector ALU: the weights of rows 0 and 1 for the lane's eight tile rows
struct SplitHeadWeights {
    float2 w[8];
};

__device__ __forceinline__ void split_head_load(SplitHeadWeights& hw, const float* __restrict__ a1, int t, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            hw.w[4 * mi + i] = *reinterpret_cast<const float2*>(a1 + ((long)t * kHk + split_image_elem(q, mi) + i) * 64 +
                                                                 32 * split_image_half(q));
}

// lacc[2 ni + o] += sum over the lane's rows of W[o][row] h[row] for window half ni
template <int NI = 2>
__device__ __forceinline__ void split_head_step(f32x16& lacc, const SplitHeadWeights& hw, const f32x16& h) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                lacc[2 * ni] = fmaf(hw.w[4 * mi + i].x, h[split_index(mi, ni, i)], lacc[2 * ni]);
                lacc[2 * ni + 1] = fmaf(hw.w[4 * mi + i].y, h[split_index(mi, ni, i)], lacc[2 * ni + 1]);
            }
}

__device__ __forceinline__ float sum_over_q(float x) {          // over the four lanes 16 q + n that share a window
    x += __shfl_xor(x, 16, 64);
    return x + __shfl_xor(x, 32, 64);
}

// (ReLU, lin2) or the bias of lin; softmax over the 2 classes; decision.  Window half ni of lane 16 q + n is
// window w_first + 16 ni; the lanes with q = 0 store.
// `unscale`: what the head's sums are multiplied by (1, or the power of two the float16 modes carry in them).
template <bool LARGE, int NI = 2>
__device__ __forceinline__ void split_head_epilogue(const f32x16& lacc, const float* __restrict__ tail, int lane, long w_first,
                                                    long n_windows, float* __restrict__ p_out, uint8_t* __restrict__ dec_out,
                                                    int mode, float threshold, float unscale = 1.0f) {
    const int q = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        float l0 = 0.0f, l1 = 0.0f;
        if (LARGE) {
            const int hh = split_image_half(q);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = split_image_elem(q, mi) + i;
                    const float v = fmaxf(fmaf(lacc[split_index(mi, ni, i)], unscale, tail[hh * 16 + e]), 0.0f);
                    l0 = fmaf(v, tail[32 + (0 * 2 + hh) * 16 + e], l0);
                    l1 = fmaf(v, tail[32 + (1 * 2 + hh) * 16 + e], l1);
                }
            l0 = sum_over_q(l0) + tail[96];
            l1 = sum_over_q(l1) + tail[97];
        } else {
            l0 = fmaf(sum_over_q(lacc[2 * ni]), unscale, tail[0]);
            l1 = fmaf(sum_over_q(lacc[2 * ni + 1]), unscale, tail[1]);
        }
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.0f / (e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        const long w = w_first + 16 * ni;
        if (q == 0 && w < n_windows) {
            *reinterpret_cast<float2*>(p_out + 2 * w) = make_float2(p0, p1);
            if (dec_out) dec_out[w] = (mode == VAD_DECIDE_THRESHOLD) ? (p1 >= threshold) : (p1 > p0);
        }
    }
}

}  // namespace rnn
}  // namespace vad
