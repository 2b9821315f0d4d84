// GRU / LSTM window classifier for gfx950: Net.forward of the reference
// (core/models.py:39-81) over many 30 x 24 windows at once.
//
// Mapping.  One wave owns a tile of 32 windows.  Every matrix product of the
// cell is computed transposed on the fp32 matrix cores,
//
//     G^T (gate-unit x window)  =  W (gate-unit x k)  .  V^T (k x window)
//
// with v_mfma_f32_32x32x2_f32: the weight matrix is the A operand (one
// 32-row tile per gate), the window data (input features or hidden state) is
// the B operand, and the 32 x 32 result has the window on the lane and the
// gate units in the 16 accumulator registers.  The result layout is the
// reason for the transposition: register i of lane half hh holds unit
// u(i, hh) = (i & 3) + 8 (i >> 2) + 4 hh, which is exactly the B-operand
// shape of the next step (k-step i takes k = 2 i + hh from lane half hh) once
// the weight columns are stored in the order u(i, hh).  The hidden state
// therefore never leaves its registers: no LDS round trip, no cross-lane
// traffic in the recurrence.  The r/z/n (or i/f/g/o) pre-activations of one
// unit land in the same lane and register index of their tiles, so the gate
// non-linearities are plain per-lane VALU work that overlaps the MFMAs of the
// other wave on the SIMD.
//
// The first Linear layer (900 -> 26, or 900 -> 2 for the small head) is
// accumulated step by step as one more 32-row tile,
// L^T += W1[:, 30 t + u] . h_t^T , so the (B, 30, 30) tensor of hidden states
// is never materialised.  fp32 MFMA is a k-ordered fmaf chain: results are
// exact fp32, no reduced-precision operand anywhere.
//
// Two kernels share this cell:
//   rnn_kernel         any window stride; the input projection W_ih x_t is part of each
//                      step's accumulation chain (100 MFMAs per GRU step).
//   rnn_stream_kernel  stride-1 windows over a packed (F, 24) matrix: a feature row is shared
//                      by up to 30 windows, so a block first projects each of its rows once
//                      (bias + W_ih x, the same fmaf chain) into LDS and the steps start their
//                      accumulators from there (64 MFMAs per GRU step).  Results are bit-identical
//                      to rnn_kernel.
// rnn_stream_split_kernel and gru_wide_split_kernel are the stride-1 kernels with the recurrent and head products
// on the bf16 matrix pipe (float32 operands split exactly into three bf16); the gru_wide_* kernels take GRUs with
// 33..96 hidden units.

#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "rnn_device.h"


namespace vad {

using namespace rnn;

// How the stride-1 GRU / LSTM kernels form their float32 matrix products (vad_set_matrix_mode).
std::atomic<int> g_matrix_mode{VAD_MATRIX_F16X6};

namespace {

// ---------------------------------------------------------------------------------------------
// any window stride

template <bool LSTM, bool LARGE, int KS>
__global__ __launch_bounds__(256, 2) void rnn_kernel(const float* __restrict__ x, long n_windows, WindowMap wm,
                                                     const float* __restrict__ blob, float* __restrict__ p_out,
                                                     uint8_t* __restrict__ dec_out, int mode, float threshold) {
    constexpr int G = LSTM ? 4 : 3;
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(LSTM, LARGE);
    for (int i = threadIdx.x; i < L.head; i += blockDim.x) lds[i] = blob[i];
    __syncthreads();
    const float* a_ih = lds + L.a_ih;
    const float* a_hh = lds + L.a_hh;
    const float* cb = lds + L.cb;
    const float* a1 = blob + L.head;                 // stays in L2

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const long tile = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long w = tile * kWindowsPerWave + col;
    if (tile * kWindowsPerWave >= n_windows) return;            // wave-uniform
    const long wc = w < n_windows ? w : n_windows - 1;          // clamp loads, mask the store
    const float* xw = window_ptr(x, wm, wc) + 12 * half;

    f32x16 h, c, lacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { h[i] = 0.0f; c[i] = 0.0f; lacc[i] = 0.0f; }

    float4 xa = *reinterpret_cast<const float4*>(xw);
    float4 xb = *reinterpret_cast<const float4*>(xw + 4);
    float4 xc = *reinterpret_cast<const float4*>(xw + 8);

    for (int t = 0; t < kT; ++t) {
        const float xv[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        {   // prefetch the next step's features
            const int tn = t + 1 < kT ? t + 1 : t;
            xa = *reinterpret_cast<const float4*>(xw + tn * kF);
            xb = *reinterpret_cast<const float4*>(xw + tn * kF + 4);
            xc = *reinterpret_cast<const float4*>(xw + tn * kF + 8);
        }
        f32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = load16(cb + (a * 2 + half) * 16);

        // input projection, K = 24: k-step s takes feature s from lane half 0 and s + 12 from half 1
#pragma unroll
        for (int s = 0; s < kKs; ++s) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA(a_ih[(g * kKs + s) * 64 + lane], xv[s], acc[g]);
        }
        // recurrent product, K = 32: k-step i takes unit u(i, half) = the lane's own h[i]; h_0 = 0
        if (t > 0) recurrent_mfma<LSTM, KS>(acc, h, a_hh, lane);
        HeadWeights<LARGE, KS> hw;                       // issued here, consumed after the gate math
        head_load<LARGE, KS>(hw, a1, t, lane);
        gate_update<LSTM>(acc, h, c);
        // head: flatten index is t * H + unit
        head_step<LARGE, KS>(lacc, hw, h);
    }
    head_finish<LARGE>(lacc);
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

// ---------------------------------------------------------------------------------------------
// stride-1 windows over a packed (F, 24) matrix, input projection hoisted per row

template <bool LSTM, int WAVES>
struct StreamGeom {
    static constexpr int G = LSTM ? 4 : 3;
    static constexpr int waves = WAVES;
    static constexpr int windows = waves * kWindowsPerWave;     // windows per block
    static constexpr int tiles = waves + 1;                     // 32-row tiles covering windows + 29 rows
    static constexpr int rs = G * 32 + 4;                       // row stride in floats: 100 / 132, keeps
                                                                // ds_read_b128 of 16 lanes on distinct banks
    static constexpr int gi_floats = tiles * 32 * rs;
};

// One block per CU (LDS-limited: the projections of 8 waves take 115 KB (GRU) / 152 KB (LSTM)), 8 waves = 2 per
// SIMD.  The recurrent A operands are loop-invariant (45 / 60 per lane): they are read from the weight blob and
// stay in registers instead of taking LDS the LSTM does not have.
template <bool LSTM>
struct StreamCfg : StreamGeom<LSTM, 8> {
    static constexpr int lds_floats = StreamGeom<LSTM, 8>::gi_floats + 4 * 2 * 16;
};

template <bool LSTM, bool LARGE, int KS>
__global__ __launch_bounds__(StreamCfg<LSTM>::waves * 64) void rnn_stream_kernel(
    const float* __restrict__ feat, long n_windows, const float* __restrict__ blob, float* __restrict__ p_out,
    uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using C = StreamCfg<LSTM>;
    constexpr int G = C::G;
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(LSTM, LARGE);
    float* gi = lds;                                             // [tiles * 32 rows][rs]
    float* cb = lds + C::gi_floats;
    for (int i = threadIdx.x; i < 4 * 2 * 16; i += C::waves * 64) cb[i] = blob[L.cb + i];
    const float* a_hh = blob + L.a_hh;

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wave = threadIdx.x >> 6;
    const long w0 = (long)blockIdx.x * C::windows;
    const long last_row = n_windows + kT - 2;                    // the matrix has n_windows + 29 rows

    // ---- phase 1: gi[row][gate * 32 + unit] = bias + W_ih x_row, one (row tile, gate) task at a time ----
    for (int task = wave; task < C::tiles * G; task += C::waves) {
        const int tile = task / G, g = task - tile * G;
        long row = w0 + tile * 32 + col;
        if (row > last_row) row = last_row;
        const float* xr = feat + row * kF + 12 * half;
        const float4 xa = *reinterpret_cast<const float4*>(xr), xb = *reinterpret_cast<const float4*>(xr + 4),
                     xc = *reinterpret_cast<const float4*>(xr + 8);
        const float xv[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        f32x16 acc = load16(blob + L.cb + (g * 2 + half) * 16);
#pragma unroll
        for (int s = 0; s < kKs; ++s) acc = MFMA(blob[L.a_ih + (g * kKs + s) * 64 + lane], xv[s], acc);
        float* dst = gi + (tile * 32 + col) * C::rs + g * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();

    // ---- phase 2: 30 steps for the wave's 32 windows ----
    const long w = w0 + wave * kWindowsPerWave + col;
    if (w0 + wave * kWindowsPerWave >= n_windows) return;       // wave-uniform, after the only barrier
    const float* a1 = blob + L.head;
    f32x16 h, c, lacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { h[i] = 0.0f; c[i] = 0.0f; lacc[i] = 0.0f; }
    const float* girow = gi + (wave * kWindowsPerWave + col) * C::rs + 4 * half;

    for (int t = 0; t < kT; ++t) {
        HeadWeights<LARGE, KS> hw;
        head_load<LARGE, KS>(hw, a1, t, lane);
        f32x16 acc[4];
        const float* src = girow + t * C::rs;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(src + g * 32 + 8 * q);
                acc[g][4 * q] = v.x; acc[g][4 * q + 1] = v.y; acc[g][4 * q + 2] = v.z; acc[g][4 * q + 3] = v.w;
            }
        if (!LSTM) acc[3] = load16(cb + (3 * 2 + half) * 16);    // b_hn
        if (t > 0) recurrent_mfma<LSTM, KS>(acc, h, a_hh, lane);
        gate_update<LSTM>(acc, h, c);
        head_step<LARGE, KS>(lacc, hw, h);
    }
    head_finish<LARGE>(lacc);
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

// ---------------------------------------------------------------------------------------------
// The same stride-1 kernel with the recurrent product and the 26-row head on the bf16 matrix pipe
// (float32 operands split exactly into three bf16, TERMS partial products per product; rnn_device.h).
// The input projection (1/30 of the matrix work) stays on the float32 MFMA.

// ni = window halves per lane.  GRU: 2, a wave owns 32 windows and 8 waves (two per SIMD) share a block's 256 windows.
// LSTM: 1, 16 windows per wave and 12 waves (three per SIMD) on 192 windows: its step is bound by vector
// instructions (10 transcendentals per unit), which a third wave per SIMD keeps issuing (measured: LSTM-small 25.0 ->
// 22.3 ms per 100 h; the GRU loses with that shape, 17.8 -> 18.4, and both lose with 16 waves of 128 registers).
template <bool LSTM>
struct SplitCfg {
    static constexpr int G = LSTM ? 4 : 3;
    static constexpr int ni = LSTM ? 1 : 2;
    static constexpr int tiles32 = LSTM ? 6 : 8;                 // 32-window groups per block
    static constexpr int waves = tiles32 * 2 / ni;
    static constexpr int wave_windows = 16 * ni;
    static constexpr int windows = tiles32 * 32;
    static constexpr int tiles = tiles32 + 1;
    // Row stride of the projections in floats.  A 16-byte read of the steps takes rows n = 0..15 from lanes with
    // q = 0 or 1 (4 floats apart): with rs / 4 = 2 (mod 16) the sixteen lanes of a read group fall on distinct
    // banks (GRU, 104).  The LSTM's LDS has no room for the next such stride (136): 132 leaves one 2-way pair.
    static constexpr int rs = LSTM ? 132 : 104;
    static constexpr int gi_floats = tiles * 32 * rs;
    static constexpr int lds_floats = gi_floats + G * kSplitTile + 4 * 2 * 16;
};

template <bool LSTM, bool LARGE, int TERMS, bool F16>
__global__ __launch_bounds__(SplitCfg<LSTM>::waves * 64) void rnn_stream_split_kernel(
    const float* __restrict__ feat, long n_windows, const float* __restrict__ blob, float* __restrict__ p_out,
    uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using C = SplitCfg<LSTM>;
    constexpr int G = C::G;
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(LSTM, LARGE);
    float* gi = lds;                                             // [tiles * 32 rows][rs]
    float* s_hh = lds + C::gi_floats;                            // [G] split tiles
    float* cb = s_hh + G * kSplitTile;
    // float16 modes: the accumulators carry 2^S (rnn_device.h); `up` = 2^S, `inv` = 2^-S from the packer
    const float up = F16 ? blob[L.tail + kTailScale(LARGE)] : 1.0f, inv = F16 ? blob[L.tail + kTailScale(LARGE) + 1] : 1.0f;
    for (int i = threadIdx.x; i < G * kSplitTile; i += C::waves * 64) s_hh[i] = blob[(F16 ? L.h_hh : L.s_hh) + i];
    for (int i = threadIdx.x; i < 4 * 2 * 16; i += C::waves * 64)
        cb[i] = blob[L.cb + i] * (F16 && !LSTM && i >= 3 * 2 * 16 ? up : 1.0f);          // b_hn starts an accumulator

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wave = threadIdx.x >> 6;
    const long w0 = (long)blockIdx.x * C::windows;
    const long last_row = n_windows + kT - 2;

    // ---- phase 1: gi[row][gate * 32 + tile row] = bias + W_ih x_row (float32 MFMA, as rnn_stream_kernel) ----
    for (int task = wave; task < C::tiles * G; task += C::waves) {
        const int tile = task / G, g = task - tile * G;
        long row = w0 + tile * 32 + col;
        if (row > last_row) row = last_row;
        const float* xr = feat + row * kF + 12 * half;
        const float4 xa = *reinterpret_cast<const float4*>(xr), xb = *reinterpret_cast<const float4*>(xr + 4),
                     xc = *reinterpret_cast<const float4*>(xr + 8);
        const float xv[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        f32x16 acc = load16(blob + L.cb + (g * 2 + half) * 16);
#pragma unroll
        for (int s = 0; s < kKs; ++s) acc = MFMA(blob[L.a_ih + (g * kKs + s) * 64 + lane], xv[s], acc);
        if (F16 && (LSTM || g < 2)) {                            // the parts a matrix product adds to: times 2^S (exact)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] *= up;
        }
        float* dst = gi + (tile * 32 + col) * C::rs + g * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();

    // ---- phase 2: 30 steps for the wave's 32 windows, in the 16 x 16 sub-tile layout (rnn_device.h): lane
    // 16 q + n holds tile rows 16 mi + 4 q + i of windows n and 16 + n.  A wave alternates between a matrix phase
    // (24 or 36 bf16 MFMAs of 16 cycles per tile) and a vector phase (gate math, bf16 split); the two waves that
    // share a SIMD (w and w + 4) overlap while they are in different phases. ----
    constexpr int NI = C::ni;
    if (w0 + wave * C::wave_windows >= n_windows) return;       // wave-uniform, after the only barrier
    const int q = lane >> 4, n = lane & 15;
    const long w_first = w0 + wave * C::wave_windows + n;        // the lane's windows: w_first (and w_first + 16)
    const float* a1 = blob + L.head;
    const u32x4* s_gate = reinterpret_cast<const u32x4*>(s_hh);
    const u32x4* s_head = reinterpret_cast<const u32x4*>(blob + (F16 ? L.h_head : L.s_head));
    f32x16 h, c, lacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { h[i] = 0.0f; c[i] = 0.0f; lacc[i] = 0.0f; }
    const float* girow = gi + (wave * C::wave_windows + n) * C::rs + 4 * q;
    const float* cb_n = cb + (3 * 2 + split_image_half(q)) * 16;             // b_hn of the lane's rows
    SplitB b;

    // One step.  kind 1: the first step (no recurrent product: the state is zero), 0: a later one, 2: either, decided by t.
    // The float16 GRU with the 2-row head peels the first step off the loop, so that its loop body has no branch around
    // the matrix work and the accumulators are not merges of two paths (GRU-small 15.5 -> 14.2 ms per 100 h).  The other
    // forms gain nothing from it or lose (same device: GRU-large f16x6 14.9 ms with the branch, 15.4 peeled; bf16x8
    // 16.6 -> 17.6) and keep the branch: what the compiler does with the freed registers decides, not the branch.
    auto step = [&](auto kind, const int t) __attribute__((always_inline)) {
        const bool with_product = decltype(kind)::value == 2 ? t > 0 : decltype(kind)::value == 0;
        SplitHeadWeights hw;                                     // 2-row head: vector ALU, float32 weights
        if (!LARGE) split_head_load(hw, a1, t, lane);
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float4 v = *reinterpret_cast<const float4*>(girow + (16 * ni + t) * C::rs + g * 32 + 16 * mi);
                    acc[g][split_index(mi, ni, 0)] = v.x; acc[g][split_index(mi, ni, 1)] = v.y;
                    acc[g][split_index(mi, ni, 2)] = v.z; acc[g][split_index(mi, ni, 3)] = v.w;
                }
        if (!LSTM) {                                             // b_hn
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float4 v = *reinterpret_cast<const float4*>(cb_n + split_image_elem(q, mi));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[3][split_index(mi, ni, 0)] = v.x; acc[3][split_index(mi, ni, 1)] = v.y;
                    acc[3][split_index(mi, ni, 2)] = v.z; acc[3][split_index(mi, ni, 3)] = v.w;
                }
            }
        }
        if (with_product) {                                      // b holds the split of h_{t-1}
            if (LARGE) lacc = split_tile_mfma<TERMS, NI, F16>(lacc, s_head + (t - 1) * (kSplitTile / 4), b, lane);
            if (LSTM) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // five tiles' operand reads hoisted together do not fit the 168 registers of a 12-wave block
                    if (LARGE) __builtin_amdgcn_sched_barrier(0);
                    acc[g] = split_tile_mfma<TERMS, NI, F16>(acc[g], s_gate + g * (kSplitTile / 4), b, lane);
                }
            } else {
                acc[0] = split_tile_mfma<TERMS, NI, F16>(acc[0], s_gate + 0 * (kSplitTile / 4), b, lane);
                acc[1] = split_tile_mfma<TERMS, NI, F16>(acc[1], s_gate + 1 * (kSplitTile / 4), b, lane);
                acc[3] = split_tile_mfma<TERMS, NI, F16>(acc[3], s_gate + 2 * (kSplitTile / 4), b, lane);
            }
        }
        if (F16) gate_update_scaled<LSTM, NI>(acc, h, c, up, inv);
        else gate_update<LSTM, NI>(acc, h, c);
        split_state<NI, F16, (TERMS > 4)>(b, h);                     // for this step's head and the next step's gates
        if (!LARGE) split_head_step<NI>(lacc, hw, h);
    };
    if (F16 && !LSTM && !LARGE) {
        step(std::integral_constant<int, 1>{}, 0);
        for (int t = 1; t < kT; ++t) step(std::integral_constant<int, 0>{}, t);
    } else {
        for (int t = 0; t < kT; ++t) step(std::integral_constant<int, 2>{}, t);
    }
    if (LARGE) lacc = split_tile_mfma<TERMS, NI, F16>(lacc, s_head + (kT - 1) * (kSplitTile / 4), b, lane);
    split_head_epilogue<LARGE, NI>(lacc, blob + L.tail, lane, w_first, n_windows, p_out, dec_out, mode, threshold,
                                   F16 ? (LARGE ? inv : kStateUnscale) : 1.0f);
}

// ---------------------------------------------------------------------------------------------
// GRU with 33..96 hidden units (UT = 2 or 3 unit tiles).  Same cell as rnn_kernel, one 32-unit tile of the new
// state at a time: the accumulators of a tile (r, z, n-input, n-hidden) see the whole previous state as B
// operand, so the new state is double-buffered in registers.  KL = k-steps of the last tile that carry units
// (13 for hidden 90); its other k-steps multiply padding and are skipped.
//   gru_wide_kernel         any window stride; the input projection is part of each step's chain.
//   gru_wide_stream_kernel  stride-1 windows: the chain's head, bias + W_ih x_row, depends on the feature row
//                           only and is shared by up to 30 windows.  row_project_kernel computes it once per row
//                           into a caller-provided workspace (the weights fill the LDS, so it cannot stay on
//                           chip as in rnn_stream_kernel) and the steps load their accumulators from there, one
//                           tile ahead of the matrix work.  Bit-identical to gru_wide_kernel.

template <int UT, int KL>
__device__ __forceinline__ constexpr int wide_ks(int kt) { return kt == UT - 1 ? KL : kHk; }

// Recurrent A operands of the wide kernels in LDS: for gate g, unit tile mt and k-tile kt the four k-steps
// 4 grp .. 4 grp + 3 of a lane are one 16-byte entry, entry index ((g UT + mt) UT 4 + kt 4 + grp) 64 + lane
// (vad_pack_net_weights writes this order for hidden > 32).  The table exceeds the 64 KB an LDS instruction's
// immediate offset spans, so it is addressed from two lane bases.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4* lds_f4_ptr;       // a 32-bit LDS address

struct WideA {
    lds_f4_ptr lo;        // entries 0 .. 63 of every lane
    lds_f4_ptr hi;        // entries 64 ..
};

__device__ __forceinline__ WideA wide_a(const float* a_hh, int lane) {
    WideA a;
    a.lo = (lds_f4_ptr)a_hh + lane;
    unsigned hi = (unsigned)(__UINTPTR_TYPE__)(a.lo + 64 * 64);
    asm volatile("" : "+v"(hi));                        // a base of its own, not lo + a constant recomputed per read
    a.hi = (lds_f4_ptr)(__UINTPTR_TYPE__)hi;
    return a;
}

template <int UT>
__device__ __forceinline__ f32x4 wide_a_entry(const WideA& a, int g, int mt, int kt, int grp) {
    const int c = (g * UT + mt) * UT * 4 + kt * 4 + grp;
    return c < 64 ? a.lo[c * 64] : a.hi[(c - 64) * 64];
}

// Step t's tile mt from its four accumulators: recurrent product, gate math.  The A operands of the next
// group of four k-steps are read while the matrix cores work on the current one.
template <int UT, int KL>
__device__ __forceinline__ void wide_tile_step(f32x16 (&acc)[4], const f32x16 (&h)[UT], f32x16& hn_mt, const f32x16& h_mt,
                                               const WideA& a, int mt, int t) {
    if (t > 0) {
        f32x4 cur[3], nxt[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) cur[g] = wide_a_entry<UT>(a, g, mt, 0, 0);
#pragma unroll
        for (int kt = 0; kt < UT; ++kt) {
            constexpr int kGroups = 4;
#pragma unroll
            for (int grp = 0; grp < kGroups; ++grp) {
                if (4 * grp >= wide_ks<UT, KL>(kt)) continue;
                // the group after this one: next group of the k-tile, or the first of the next k-tile
                const bool more_here = 4 * (grp + 1) < wide_ks<UT, KL>(kt);
                if (more_here || kt + 1 < UT) {
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        nxt[g] = wide_a_entry<UT>(a, g, mt, more_here ? kt : kt + 1, more_here ? grp + 1 : 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // the reads above go out before this group's matrix work
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * grp + q;
                    if (i >= wide_ks<UT, KL>(kt)) continue;
                    const float hv = h[kt][i];
                    acc[0] = MFMA(cur[0][q], hv, acc[0]);
                    acc[1] = MFMA(cur[1][q], hv, acc[1]);
                    acc[3] = MFMA(cur[2][q], hv, acc[3]);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) cur[g] = nxt[g];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float r = sigmoid_pre(acc[0][i]);
        const float z = sigmoid_pre(acc[1][i]);
        const float n = tanh_pre(fmaf(r, acc[3][i], acc[2][i]));
        hn_mt[i] = fmaf(z, h_mt[i] - n, n);
    }
}

// The step-t slice of the 26-row (or 2-row) head for the wide models.  Its A operands come from the weight
// blob in global memory (L2): all of the step's loads are issued before the first product waits for one (left
// to itself the compiler issues them one at a time, each behind the product that uses the previous one).
template <int UT, int KL>
__device__ __forceinline__ void wide_head_step(f32x16& lacc, const f32x16 (&h)[UT], const float* __restrict__ a1, int t, int lane) {
    constexpr int kSteps = (UT - 1) * kHk + KL;
    float aw[kSteps];
    {
        int j = 0;
#pragma unroll
        for (int kt = 0; kt < UT; ++kt)
#pragma unroll
            for (int i = 0; i < wide_ks<UT, KL>(kt); ++i) aw[j++] = a1[(t * UT * kHk + kt * kHk + i) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        int j = 0;
#pragma unroll
        for (int kt = 0; kt < UT; ++kt)
#pragma unroll
            for (int i = 0; i < wide_ks<UT, KL>(kt); ++i) lacc = MFMA(aw[j++], h[kt][i], lacc);
    }
}

template <bool LARGE, int UT, int KL>
__global__ __launch_bounds__(256, 1) void gru_wide_kernel(const float* __restrict__ x, long n_windows, WindowMap wm,
                                                          const float* __restrict__ blob, float* __restrict__ p_out,
                                                          uint8_t* __restrict__ dec_out, int mode, float threshold) {
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(false, LARGE, UT);
    for (int i = threadIdx.x; i < L.head; i += blockDim.x) lds[i] = blob[i];
    __syncthreads();
    const float* a_ih = lds + L.a_ih;
    const float* a_hh = lds + L.a_hh;
    const float* cb = lds + L.cb;
    const float* a1 = blob + L.head;

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const long tile = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long w = tile * kWindowsPerWave + col;
    if (tile * kWindowsPerWave >= n_windows) return;
    const long wc = w < n_windows ? w : n_windows - 1;
    const float* xw = window_ptr(x, wm, wc) + 12 * half;
    const WideA wa = wide_a(a_hh, lane);

    f32x16 h[UT], hn[UT], lacc;
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) h[u][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.0f;

    for (int t = 0; t < kT; ++t) {
        const float4 xa = *reinterpret_cast<const float4*>(xw + t * kF);
        const float4 xb = *reinterpret_cast<const float4*>(xw + t * kF + 4);
        const float4 xc = *reinterpret_cast<const float4*>(xw + t * kF + 8);
        const float xv[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
#pragma unroll
        for (int mt = 0; mt < UT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);           // keep the A-operand reads of later tiles from piling up
            f32x16 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = load16(cb + ((a * UT + mt) * 2 + half) * 16);
#pragma unroll
            for (int sft = 0; sft < kKs; ++sft)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a_ih[((g * UT + mt) * kKs + sft) * 64 + lane], xv[sft], acc[g]);
            wide_tile_step<UT, KL>(acc, h, hn[mt], h[mt], wa, mt, t);
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) h[u] = hn[u];
        __builtin_amdgcn_sched_barrier(0);
        wide_head_step<UT, KL>(lacc, h, a1, t, lane);
    }
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

// gi[row][(g * UT + mt) * 32 + half * 16 + i] = accumulator register i of lane (row, half) after bias + W_ih x_row
// for gate g (r, z, n-input), unit tile mt: exactly the registers a step of the stream kernel starts from.
// scale_at >= 0: the float16 modes' 2^S sits at blob[scale_at] and multiplies the r and z parts, which the recurrent
// products add to (the n-input part is only ever read by the gate math).
template <int UT>
__global__ __launch_bounds__(256) void row_project_kernel(const float* __restrict__ feat, long n_rows,
                                                          const float* __restrict__ blob, float* __restrict__ gi,
                                                          int scale_at) {
    const NetLayout L = net_layout(false, false, UT);            // a_ih and cb do not depend on the head
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + col;
    if (row - col >= n_rows) return;                             // wave-uniform
    const long rc = row < n_rows ? row : n_rows - 1;
    const float* xr = feat + rc * kF + 12 * half;
    const float4 xa = *reinterpret_cast<const float4*>(xr), xb = *reinterpret_cast<const float4*>(xr + 4),
                 xc = *reinterpret_cast<const float4*>(xr + 8);
    const float xv[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
    float* dst = gi + rc * (3 * UT * 32) + half * 16;
#pragma unroll
    for (int gm = 0; gm < 3 * UT; ++gm) {
        f32x16 acc = load16(blob + L.cb + (gm * 2 + half) * 16);
#pragma unroll
        for (int s = 0; s < kKs; ++s) acc = MFMA(blob[L.a_ih + (gm * kKs + s) * 64 + lane], xv[s], acc);
        if (scale_at >= 0 && gm < 2 * UT) {
            const float up = blob[scale_at];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] *= up;
        }
        if (row < n_rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dst + gm * 32 + 4 * q) =
                    make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// Lane l takes the value of lane l + 1 (whole-wave shift by one lane, no LDS traffic); lane 63 keeps its own.
__device__ __forceinline__ float lane_up(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}

// A lane (window column `col`, half) holds the projected row its window consumes at the current step in
// registers, 3 * UT tiles of 16.  Window w + 1 consumes at step t the row that window w consumes at step t + 1,
// so from one step to the next the rows move down by one lane and only the wave's last column loads a new
// one: the workspace is read once per row and wave (plus 29 rows of overlap per 32 windows) instead of 30 times.
template <bool LARGE, int UT, int KL>
__global__ __launch_bounds__(256, 1) void gru_wide_stream_kernel(const float* __restrict__ gi, long n_windows,
                                                                 const float* __restrict__ blob, float* __restrict__ p_out,
                                                                 uint8_t* __restrict__ dec_out, int mode, float threshold) {
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(false, LARGE, UT);
    // the recurrent A operands and the biases (the input projection's operands are not needed here)
    for (int i = L.a_hh + threadIdx.x; i < L.head; i += blockDim.x) lds[i - L.a_hh] = blob[i];
    __syncthreads();
    const float* a_hh = lds;
    const float* cb = lds + (L.cb - L.a_hh);
    const float* a1 = blob + L.head;

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const long tile = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long w = tile * kWindowsPerWave + col;
    if (tile * kWindowsPerWave >= n_windows) return;
    const long last_row = n_windows + kT - 2;                    // the workspace holds n_windows + 29 rows
    constexpr int kRow = 3 * UT * 32;                            // floats per projected row
    const WideA wa = wide_a(a_hh, lane);

    f32x16 h[UT], hn[UT], lacc, row[3 * UT];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) h[u][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.0f;
    {
        const float* src = gi + (w < last_row ? w : last_row) * kRow + half * 16;      // step 0: row w
#pragma unroll
        for (int gm = 0; gm < 3 * UT; ++gm) row[gm] = load16(src + gm * 32);
    }

    for (int t = 0; t < kT; ++t) {
        // the row the wave's last column consumes at the next step
        const long rn = tile * kWindowsPerWave + 31 + t + 1;
        const float* nxt = gi + (rn < last_row ? rn : last_row) * kRow + half * 16;
#pragma unroll
        for (int mt = 0; mt < UT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = row[g * UT + mt];
            acc[3] = load16(cb + ((3 * UT + mt) * 2 + half) * 16);
            wide_tile_step<UT, KL>(acc, h, hn[mt], h[mt], wa, mt, t);
            if (t + 1 < kT) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    f32x16 v;
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = lane_up(row[g * UT + mt][i]);
                    if (col == 31) v = load16(nxt + (g * UT + mt) * 32);       // in flight until the next step
                    row[g * UT + mt] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) h[u] = hn[u];
        __builtin_amdgcn_sched_barrier(0);
        wide_head_step<UT, KL>(lacc, h, a1, t, lane);
    }
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

// ---------------------------------------------------------------------------------------------
// The stride-1 wide GRU with the recurrent product and the head on the bf16 matrix pipe (float32 operands split
// exactly into three bf16, rnn_device.h), in the 16 x 16 sub-tile layout of rnn_stream_split_kernel: lane 16 q + n
// holds, of every 32-unit tile, tile rows 16 mi + 4 q + i of windows n and 16 + n.  The recurrent products read the
// split copy of the state, so a tile's new state overwrites the old one in place.  The split A operands of all
// gates fill the LDS except the last gate's lo terms, which come from the weight blob (L2) like the head's.
// Projected rows come from row_project_kernel's workspace and move down one lane per step inside each 16-lane row
// (the row of window 15 comes from the lane's second window, that of window 31 is loaded).

// NI = window halves per lane: 2 -> a wave owns 32 windows and runs alone on its SIMD (512 registers), 1 -> 16
// windows per wave, two waves per SIMD whose matrix and vector phases overlap.
template <int UT, int NI>
struct WideSplitB {
    u32x4 hi[UT][NI], mid[UT][NI], lo[UT][NI];      // [k-block][window half]
};

// state of one 32-unit tile: [mi][ni] groups of the lane's four tile rows
template <int NI>
struct WideTile {
    f32x4s v[2][NI];
};

template <int UT, int NI, bool F16, bool WITH_LO>
__device__ __forceinline__ void wide_split_state(WideSplitB<UT, NI>& b, const WideTile<NI> (&h)[UT]) {
#pragma unroll
    for (int kb = 0; kb < UT; ++kb)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {                            // k-slots 2 jp, 2 jp + 1 = rows (jp >> 1, 2 (jp & 1) ..)
                const SplitTerms s = split_pair<F16, WITH_LO>(h[kb].v[jp >> 1][ni][2 * (jp & 1)], h[kb].v[jp >> 1][ni][2 * (jp & 1) + 1]);
                b.hi[kb][ni][jp] = s.hi;
                b.mid[kb][ni][jp] = s.mid;
                b.lo[kb][ni][jp] = s.lo;
            }
}

// c[ni] += A (hi, mid, lo) . B (k-block kb): the TERMS partial products, smallest first
template <int UT, int NI, int TERMS, bool F16>
__device__ __forceinline__ void wide_split_terms(f32x4s (&c)[NI], const u32x4& ah, const u32x4& am, const u32x4& al,
                                                 const WideSplitB<UT, NI>& b, int kb) {
    auto term = [&](const u32x4& at, const u32x4 (&bt)[NI]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) c[ni] = mfma_split<F16>(at, bt[ni], c[ni]);
    };
    if (TERMS == 9) term(al, b.lo[kb]);
    if (TERMS >= 8) {
        term(am, b.lo[kb]);
        term(al, b.mid[kb]);
    }
    if (TERMS >= 6) {
        term(ah, b.lo[kb]);
        term(al, b.hi[kb]);
    }
    term(am, b.mid[kb]);
    term(ah, b.mid[kb]);
    term(am, b.hi[kb]);
    term(ah, b.hi[kb]);
}

// Row registers of one (gate, tile, mi, i): lanes 0..14 of a 16-lane row take the value of the next lane; lane 15
// of the first window half takes lane 0 of the second half's register, and the last half's lane 15 is left for the
// caller to load.
template <int NI>
__device__ __forceinline__ void rows_down(f32x4s (&x)[NI], int i) {
    if (NI == 2) {
        const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x[NI - 1][i]), 0x12F /* row_ror:15 */, 0xF, 0xF, false);
        x[0][i] = __int_as_float(__builtin_amdgcn_update_dpp(r, __float_as_int(x[0][i]), 0x101 /* row_shl:1 */, 0xF, 0xF, false));
        x[NI - 1][i] = __int_as_float(r);
    } else {
        const int v = __float_as_int(x[0][i]);
        x[0][i] = __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x101 /* row_shl:1 */, 0xF, 0xF, false));
    }
}

// The split A operands in LDS: up to 158 entries of 1 KB, more than the 64 KB an LDS instruction's immediate offset
// spans.  Three lane bases 64 entries apart, each opaque to the compiler: from a single pointer it forms one
// address register per distant entry, hoists them all out of the step loop and spills them -- and every reload from
// scratch waits on the vector-memory counter, i.e. for the projected-row loads still in flight from the last step.
typedef const __attribute__((address_space(3))) u32x4* lds_u4_ptr;

struct WideSplitA {
    lds_u4_ptr b[3];
};

__device__ __forceinline__ WideSplitA wide_split_a(const float* lds, int lane) {
    WideSplitA a;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        unsigned at = (unsigned)(__UINTPTR_TYPE__)((lds_u4_ptr)lds + lane + k * 64 * 64);
        asm volatile("" : "+v"(at));
        a.b[k] = (lds_u4_ptr)(__UINTPTR_TYPE__)at;
    }
    return a;
}

__device__ __forceinline__ u32x4 wide_split_entry(const WideSplitA& a, int e) {
    return e < 64 ? a.b[0][e * 64] : (e < 128 ? a.b[1][(e - 64) * 64] : a.b[2][(e - 128) * 64]);
}

template <int NI>
struct WideSplitCfg {
    static constexpr int waves = 8 / NI;                         // per block: one or two per SIMD
    static constexpr int wave_windows = 16 * NI;
};

// Diagnostic build only (-DVAD_TIMELINE=<block>): s_memtime stamps of the two waves that share SIMD 0 of one block, at
// the phase boundaries of every step, read back through vad_debug_timeline (tools/timeline_wide.py).
#if defined(VAD_TIMELINE)
constexpr int kTlStamps = 16;
__device__ unsigned long long g_timeline[2][kT][kTlStamps];
#define TL_STAMP(k)                                                                                       \
    do {                                                                                                  \
        const unsigned long long tl_now = __builtin_amdgcn_s_memtime();                                   \
        if (blockIdx.x == VAD_TIMELINE && (threadIdx.x & 255) == 0) g_timeline[threadIdx.x >> 8][t][k] = tl_now; \
    } while (0)
#else
#define TL_STAMP(k)
#endif

template <bool LARGE, int UT, int TERMS, int NI, bool F16>
__global__ __launch_bounds__(WideSplitCfg<NI>::waves * 64) void gru_wide_split_kernel(
    const float* __restrict__ gi, long n_windows, const float* __restrict__ blob, float* __restrict__ p_out,
    uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using C = WideSplitCfg<NI>;
    extern __shared__ __align__(16) float lds[];
    const NetLayout L = net_layout(false, LARGE, UT);
    constexpr int kHm = wide_split_hm_entries(UT);
    constexpr int kLdsEntries = TERMS > 4 ? wide_split_lds_entries(UT) : kHm;      // four products: no lo terms
    const u32x4* a_glob = reinterpret_cast<const u32x4*>(blob + (F16 ? L.h_hh : L.s_hh));
    {
        u32x4* dst = reinterpret_cast<u32x4*>(lds);
        for (int i = threadIdx.x; i < kLdsEntries * 64; i += blockDim.x) dst[i] = a_glob[i];
        float* cbw = lds + kLdsEntries * kSplitEntry;
        for (int i = threadIdx.x; i < 4 * UT * 32; i += blockDim.x)
            cbw[i] = blob[L.cb + i] * (F16 && i >= 3 * UT * 32 ? blob[L.tail + kTailScale(LARGE)] : 1.0f);   // b_hn
    }
    __syncthreads();
    const float* cb = lds + kLdsEntries * kSplitEntry;
    const u32x4* a_head = reinterpret_cast<const u32x4*>(blob + (F16 ? L.h_head : L.s_head));
    const float* a1 = blob + L.head;

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, n = lane & 15;
    const long tile = (long)blockIdx.x * C::waves + (threadIdx.x >> 6);
    if (tile * C::wave_windows >= n_windows) return;             // wave-uniform, after the only barrier
    const float up = F16 ? blob[L.tail + kTailScale(LARGE)] : 1.0f, inv = F16 ? blob[L.tail + kTailScale(LARGE) + 1] : 1.0f;
    const WideSplitA a_lds = wide_split_a(lds, lane);
    const long w_first = tile * C::wave_windows + n;             // the lane's windows: w_first (and w_first + 16)
    const long last_row = n_windows + kT - 2;                    // the workspace holds n_windows + 29 rows
    constexpr int kRow = 3 * UT * 32;                            // floats per projected row
    // tile rows 16 mi + 4 q + i of a 32-row image stored per lane half (hh, element): hh = q & 1, elements
    // 4 (2 mi + (q >> 1)) + i, i.e. float offset 16 hh + that
    const int chunk = 16 * split_image_half(q) + split_image_elem(q, 0);            // mi = 0; mi = 1 is 8 floats on
    // One opaque base each for the lane's b_hn biases in LDS and for the operand entries that stay in global memory: left
    // to itself the compiler keeps one address register (pair) per tile and entry alive across the step loop.
    typedef const __attribute__((address_space(3))) f32x4s* lds_f4s_ptr;
    unsigned cb_at = (unsigned)(__UINTPTR_TYPE__)((const __attribute__((address_space(3))) float*)(cb + 3 * UT * 32 + chunk));
    asm volatile("" : "+v"(cb_at));
    typedef const __attribute__((address_space(1))) u32x4* global_u4_ptr;
    global_u4_ptr far_base = (global_u4_ptr)(a_glob + (size_t)kLdsEntries * 64 + lane);
    asm volatile("" : "+v"(far_base));

    WideTile<NI> h[UT], lacc, row[3 * UT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int u = 0; u < UT; ++u) h[u].v[mi][ni] = f32x4s{0.0f, 0.0f, 0.0f, 0.0f};
            lacc.v[mi][ni] = f32x4s{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {                            // step 0: rows w_first (and w_first + 16)
        const long r0 = w_first + 16 * ni;
        const float* src = gi + (r0 < last_row ? r0 : last_row) * kRow + chunk;
#pragma unroll
        for (int gm = 0; gm < 3 * UT; ++gm)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) row[gm].v[mi][ni] = *reinterpret_cast<const f32x4s*>(src + gm * 32 + 8 * mi);
    }
    WideSplitB<UT, NI> b;
    float head2[NI][2];                                          // 2-row head: running sums of the lane's rows
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) head2[ni][0] = head2[ni][1] = 0.0f;

    // One step.  The first step has no recurrent product (the state is zero); it is peeled off the loop so that the
    // loop body has no branch around its matrix work: with one, every accumulator is a merge of two paths, and the
    // copies and the registers that costs stay in the hot loop (the float16 form spilled 15 registers and needed 116 ms
    // per 100 h for hidden 90; without the branch 0 and 100 ms).  The row requests are unconditional for the same
    // reason (past the last step they re-read the last row, clamped, and nothing uses it).
    auto step = [&](auto first_step, const int t) __attribute__((always_inline)) {
        constexpr bool kFirst = decltype(first_step)::value;
        // the row the wave's last window consumes at the next step
        const long rn = tile * C::wave_windows + (C::wave_windows - 1) + t + 1;
        const float* nxt = gi + (rn < last_row ? rn : last_row) * kRow + chunk;
        // The rows of tile mt move down a lane and the last window's next row is requested: in flight until the next step.
        auto next_rows = [&](int mt) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                WideTile<NI>& rv = row[g * UT + mt];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) rows_down<NI>(rv.v[mi], i);
                if (n == 15) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        rv.v[mi][NI - 1] = *reinterpret_cast<const f32x4s*>(nxt + (g * UT + mt) * 32 + 8 * mi);
                }
            }
        };
        TL_STAMP(0);
#pragma unroll
        for (int mt = 0; mt < UT; ++mt) {
            WideTile<NI> acc[4];
            // The tile's 6 UT operand units (gate, row group, k-block: a hi, mid and lo entry, TERMS products) stream through
            // a ring of three: unit u + 2 is requested from LDS before unit u's products are issued, so the reads of two
            // units are in flight behind the matrix work and no product waits for a whole LDS round trip (left to the
            // compiler, a wave drained its LDS queue about every six products: hidden 90 100 -> 94 ms per 100 h).
            struct AUnit {
                u32x4 h, m, l;
            };
            constexpr int kRing = 3, kUnits = 6 * UT;
            AUnit ring[kRing];
            auto fetch = [&](int u, AUnit& o) {
                const int base = wide_split_base(UT, u / (2 * UT), mt, (u / UT) & 1, u % UT);
                o.h = wide_split_entry(a_lds, base * 2 + 0);
                o.m = wide_split_entry(a_lds, base * 2 + 1);
                if (TERMS > 4 && kHm + base < kLdsEntries) o.l = wide_split_entry(a_lds, kHm + base);
            };
            if (!kFirst) {
#pragma unroll
                for (int u = 0; u + 1 < kRing; ++u) fetch(u, ring[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = row[g * UT + mt];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {                     // b_hn
                const f32x4s v = *(lds_f4s_ptr)(__UINTPTR_TYPE__)(cb_at + (mt * 32 + 8 * mi) * 4);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[3].v[mi][ni] = v;
            }
            if (!kFirst) {
                u32x4 far[2][UT];
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    const int g = u / (2 * UT), mi = (u / UT) & 1, kb = u % UT;
                    if (u == 2 * UT) {
                        // the entries that LDS has no room for (the tail of the last gate's lo terms): requested from the
                        // weight blob a gate ahead of the one that uses them
#pragma unroll
                        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                            for (int fk = 0; fk < UT; ++fk) {
                                const int e = kHm + wide_split_base(UT, 2, mt, fm, fk);
                                if (TERMS > 4 && e >= kLdsEntries) far[fm][fk] = far_base[(e - kLdsEntries) * 64];
                            }
                    }
                    if (u + kRing - 1 < kUnits) fetch(u + kRing - 1, ring[(u + kRing - 1) % kRing]);
                    __builtin_amdgcn_sched_barrier(0);           // the reads above go out before this unit's products
                    const AUnit& o = ring[u % kRing];
                    const bool is_far = TERMS > 4 && kHm + wide_split_base(UT, g, mt, mi, kb) >= kLdsEntries;
                    wide_split_terms<UT, NI, TERMS, F16>(acc[g == 2 ? 3 : g].v[mi], o.h, o.m,
                                                         TERMS > 4 ? (is_far ? far[mi][kb] : o.l) : o.h, b, kb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            TL_STAMP(1 + 4 * mt);                               // matrix work issued
#if defined(VAD_TIMELINE)
            asm volatile("" : "+v"(acc[0].v[0][0]), "+v"(acc[1].v[0][0]), "+v"(acc[3].v[1][0]));   // results have arrived
            TL_STAMP(2 + 4 * mt);
#endif
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float z, nn;
                        if (F16) {                               // accumulators times 2^S, state times 2^14 (rnn_device.h)
                            const float r_inv = __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(acc[0].v[mi][ni][i] * inv), up, up));
                            z = sigmoid_pre(acc[1].v[mi][ni][i] * inv);
                            nn = fmaf(-2.0f * kStateScale, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(
                                          fmaf(r_inv, acc[3].v[mi][ni][i], acc[2].v[mi][ni][i]))), kStateScale);
                        } else {
                            const float r = sigmoid_pre(acc[0].v[mi][ni][i]);
                            z = sigmoid_pre(acc[1].v[mi][ni][i]);
                            nn = tanh_pre(fmaf(r, acc[3].v[mi][ni][i], acc[2].v[mi][ni][i]));
                        }
                        h[mt].v[mi][ni][i] = fmaf(z, h[mt].v[mi][ni][i] - nn, nn);   // in place: the products read the split copy
                    }
            // The last tile's request waits until the head is done: the head's operands come from global memory too, the
            // memory counter retires in order, and waiting for them right after this request would mean waiting for a
            // row that is not needed for another two thirds of a step.
            TL_STAMP(3 + 4 * mt);                               // gate math done
            if (mt + 1 < UT) next_rows(mt);
            TL_STAMP(4 + 4 * mt);
        }
        if (F16) __builtin_amdgcn_sched_barrier(0);              // (the conversions interleaved with the gate math spill)
        // The 26-row head's operands of this step come from the weight blob (L2): a row group's nine entries are requested
        // at once, in registers the gate accumulators have just left and before the state is split, so the step waits
        // for two round trips instead of one per k-block (all eighteen at once need 72 registers and gain nothing more).
        constexpr int kHeadTerms = TERMS > 4 ? 3 : 2;
        u32x4 he[UT][3];
        auto head_request = [&](int mi) {
#pragma unroll
            for (int kb = 0; kb < UT; ++kb) {
                const u32x4* e = a_head + (size_t)(((t * 2 + mi) * UT + kb) * 3) * 64 + lane;
#pragma unroll
                for (int k = 0; k < kHeadTerms; ++k) he[kb][k] = e[64 * k];
            }
        };
        SplitHeadWeights hw[UT];                                 // 2-row head: vector ALU, float32 weights
        if (LARGE) {
            head_request(0);
        } else {
#pragma unroll
            for (int kb = 0; kb < UT; ++kb) split_head_load(hw[kb], a1 + kb * kHk * 64, t * UT, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        wide_split_state<UT, NI, F16, (TERMS > 4)>(b, h);         // for this step's head and the next step's gates
        TL_STAMP(13);
        if (LARGE) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kb = 0; kb < UT; ++kb)
                    wide_split_terms<UT, NI, TERMS, F16>(lacc.v[mi], he[kb][0], he[kb][1], he[kb][TERMS > 4 ? 2 : 0], b, kb);
                __builtin_amdgcn_sched_barrier(0);
                if (mi == 0) head_request(1);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < UT; ++kb)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            head2[ni][0] = fmaf(hw[kb].w[4 * mi + i].x, h[kb].v[mi][ni][i], head2[ni][0]);
                            head2[ni][1] = fmaf(hw[kb].w[4 * mi + i].y, h[kb].v[mi][ni][i], head2[ni][1]);
                        }
        }
        __builtin_amdgcn_sched_barrier(0);                       // keep the request below the head's waits
        TL_STAMP(14);
        next_rows(UT - 1);
        TL_STAMP(15);
    };
    step(std::true_type{}, 0);
    for (int t = 1; t < kT; ++t) step(std::false_type{}, t);
    // epilogue in the layout split_head_epilogue reads (element (mi, ni, i) = split_index; 2-row head: elements 2 ni, 2 ni + 1)
    f32x16 out;
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = 0.0f;
    if (LARGE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i) out[split_index(mi, ni, i)] = lacc.v[mi][ni][i];
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            out[2 * ni] = head2[ni][0];
            out[2 * ni + 1] = head2[ni][1];
        }
    }
    split_head_epilogue<LARGE, NI>(out, blob + L.tail, lane, w_first, n_windows, p_out, dec_out, mode, threshold,
                                   F16 ? (LARGE ? inv : kStateUnscale) : 1.0f);
}

// The split matrix modes as compile-time (partial products, float16 operands) pairs.
constexpr int kSplitModes[] = {VAD_MATRIX_BF16X6, VAD_MATRIX_BF16X9, VAD_MATRIX_BF16X8, VAD_MATRIX_F16X6, VAD_MATRIX_F16X4};

template <typename F>
void dispatch_split_mode(int matrix_mode, F&& f) {
    using std::integral_constant;
    switch (matrix_mode) {
        case VAD_MATRIX_BF16X9: f(integral_constant<int, 9>{}, std::false_type{}); break;
        case VAD_MATRIX_BF16X8: f(integral_constant<int, 8>{}, std::false_type{}); break;
        case VAD_MATRIX_BF16X6: f(integral_constant<int, 6>{}, std::false_type{}); break;
        case VAD_MATRIX_F16X6: f(integral_constant<int, 6>{}, std::true_type{}); break;
        default: f(integral_constant<int, 4>{}, std::true_type{}); break;      // VAD_MATRIX_F16X4
    }
}

template <typename K>
int allow_lds(K kernel, int floats) {
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      floats * (int)sizeof(float)));
    return VAD_OK;
}

template <bool LARGE, int UT, int KL>
int allow_wide() {
    const NetLayout L = net_layout(false, LARGE, UT);
    int rc = allow_lds(&gru_wide_kernel<LARGE, UT, KL>, L.head);
    if (rc != VAD_OK) return rc;
    return allow_lds(&gru_wide_stream_kernel<LARGE, UT, KL>, L.head - L.a_hh);
}

constexpr int wide_split_lds_floats(int ut, int terms) {
    return (terms > 4 ? wide_split_lds_entries(ut) : wide_split_hm_entries(ut)) * kSplitEntry + 4 * ut * 32;
}

constexpr int kWideSplitNi = 1;       // window halves per lane of gru_wide_split_kernel: 16 windows per wave, two waves per SIMD
                                      // (2 = 32 windows per wave, one wave per SIMD with all 512 registers: measured 199 vs 148 ms)

// The 2-row head on 65..96 units (no model of the reference has it) does not fit 256 registers in the bfloat16 forms
// (more than 100 spilled): those take the float32 workspace kernel, which is what the spilling kernel's time came to.
template <bool LARGE, int UT, bool F16>
constexpr bool wide_split_built() { return LARGE || UT < 3 || F16; }

template <bool LARGE, int UT>
int allow_wide_split() {
    int rc = VAD_OK;
    for (int mode : kSplitModes)
        dispatch_split_mode(mode, [&](auto terms, auto f16) {
            if constexpr (wide_split_built<LARGE, UT, decltype(f16)::value>()) {
                if (rc == VAD_OK)
                    rc = allow_lds(&gru_wide_split_kernel<LARGE, UT, decltype(terms)::value, kWideSplitNi, decltype(f16)::value>,
                                   wide_split_lds_floats(UT, decltype(terms)::value));
            }
        });
    return rc;
}

template <bool LSTM, bool LARGE>
int allow_stream_split() {
    int rc = VAD_OK;
    for (int mode : kSplitModes)
        dispatch_split_mode(mode, [&](auto terms, auto f16) {
            if (rc == VAD_OK)
                rc = allow_lds(&rnn_stream_split_kernel<LSTM, LARGE, decltype(terms)::value, decltype(f16)::value>,
                               SplitCfg<LSTM>::lds_floats);
        });
    return rc;
}

}  // namespace

// Kernels that use more than 64 KB of dynamic LDS must be opted in once per device.
int ensure_rnn_attrs() {
    static std::mutex mu;                       // callers on different host threads may race to the first launch
    static bool done[64] = {false};
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return VAD_OK;
    int rc = VAD_OK;
    if ((rc = allow_lds(&rnn_stream_kernel<false, false, 15>, StreamCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<false, true, 15>, StreamCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<true, false, 15>, StreamCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<true, true, 15>, StreamCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<false, false, 16>, StreamCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<false, true, 16>, StreamCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<true, false, 16>, StreamCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_kernel<true, true, 16>, StreamCfg<true>::lds_floats))) return rc;
    if ((rc = allow_stream_split<false, false>()) || (rc = allow_stream_split<false, true>()) ||
        (rc = allow_stream_split<true, false>()) || (rc = allow_stream_split<true, true>()))
        return rc;
    if ((rc = allow_wide<false, 2, 13>()) || (rc = allow_wide<true, 2, 13>()) || (rc = allow_wide<false, 3, 13>()) ||
        (rc = allow_wide<true, 3, 13>()) || (rc = allow_wide<false, 2, 16>()) || (rc = allow_wide<true, 2, 16>()) ||
        (rc = allow_wide<false, 3, 16>()) || (rc = allow_wide<true, 3, 16>()))
        return rc;
    if ((rc = allow_wide_split<false, 2>()) || (rc = allow_wide_split<true, 2>()) || (rc = allow_wide_split<false, 3>()) ||
        (rc = allow_wide_split<true, 3>()))
        return rc;
    done[dev] = true;
    return VAD_OK;
}

namespace {

// Scratch of the stride-1 wide GRU: one projected row (3 gates x UT tiles x 32 floats) per feature row of a chunk.
struct Workspace {
    float* ptr;
    size_t bytes;
};

template <bool LARGE, int UT, int KL>
int launch_wide(const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec, int mode,
                float threshold, Workspace ws, int matrix_mode, hipStream_t st) {
    const NetLayout L = net_layout(false, LARGE, UT);
    int rc = ensure_rnn_attrs();
    if (rc != VAD_OK) return rc;
    const long per_block = (long)kWindowsPerWave * kWavesPerBlock;
    const size_t row_bytes = (size_t)3 * UT * 32 * sizeof(float);
    const long ws_rows = ws.ptr ? (long)(ws.bytes / row_bytes) : 0;
    if (!wm.rows && wm.stride == kF && ws_rows >= kT) {
        // stride-1 windows: chunks of as many windows as the workspace has rows for; the projection of a chunk's
        // rows, then its windows (stream order makes the rows visible to the second kernel and keeps the next
        // chunk's projection from overwriting rows still in use)
        const long chunk = ws_rows - (kT - 1);
        const size_t lds_bytes = (size_t)(L.head - L.a_hh) * sizeof(float);
        for (long w0 = 0; w0 < n_windows; w0 += chunk) {
            const long nw = n_windows - w0 < chunk ? n_windows - w0 : chunk;
            const long rows = nw + kT - 1;
            const long pblocks = (rows + 127) / 128, blocks = (nw + per_block - 1) / per_block;
            VAD_REQUIRE(pblocks <= 0x7FFFFFFFL && blocks <= 0x7FFFFFFFL, "too many windows for one launch");
            const bool f16 = matrix_mode == VAD_MATRIX_F16X6 || matrix_mode == VAD_MATRIX_F16X4;
            hipLaunchKernelGGL((row_project_kernel<UT>), dim3((unsigned)pblocks), dim3(256), 0, st, x + w0 * kF, rows, blob,
                               ws.ptr, f16 ? L.tail + kTailScale(LARGE) : -1);
            using SC = WideSplitCfg<kWideSplitNi>;
            const long sblocks = (nw + SC::waves * SC::wave_windows - 1) / (SC::waves * SC::wave_windows);
            if (matrix_mode == VAD_MATRIX_F32 || !(f16 ? wide_split_built<LARGE, UT, true>() : wide_split_built<LARGE, UT, false>()))
                hipLaunchKernelGGL((gru_wide_stream_kernel<LARGE, UT, KL>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock),
                                   lds_bytes, st, ws.ptr, nw, blob, p + 2 * w0, dec ? dec + w0 : nullptr, mode, threshold);
            else
                dispatch_split_mode(matrix_mode, [&](auto terms, auto f16_mode) {
                    constexpr int kTerms = decltype(terms)::value;
                    if constexpr (wide_split_built<LARGE, UT, decltype(f16_mode)::value>())
                        hipLaunchKernelGGL((gru_wide_split_kernel<LARGE, UT, kTerms, kWideSplitNi, decltype(f16_mode)::value>),
                                           dim3((unsigned)sblocks), dim3(64 * SC::waves),
                                           (size_t)wide_split_lds_floats(UT, kTerms) * sizeof(float), st, ws.ptr, nw, blob,
                                           p + 2 * w0, dec ? dec + w0 : nullptr, mode, threshold);
                });
            VAD_HIP_CHECK(hipGetLastError());
        }
        return VAD_OK;
    }
    const size_t lds_bytes = (size_t)L.head * sizeof(float);
    const long blocks = (n_windows + per_block - 1) / per_block;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
    hipLaunchKernelGGL((gru_wide_kernel<LARGE, UT, KL>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds_bytes, st, x,
                       n_windows, wm, blob, p, dec, mode, threshold);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

template <bool LARGE, int UT>
int launch_wide_kl(int hidden, const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec,
                   int mode, float threshold, Workspace ws, int matrix_mode, hipStream_t st) {
    // k-steps of the last tile that carry units; the two instantiations cover hidden 90 (13) and everything else
    return tile_k_steps(tile_units(hidden, UT - 1)) <= 13
               ? launch_wide<LARGE, UT, 13>(x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st)
               : launch_wide<LARGE, UT, 16>(x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st);
}

// ---------------------------------------------------------------------------------------------
// launchers

template <bool LSTM, bool LARGE, int KS>
int launch_ks(const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec, int mode,
           float threshold, int matrix_mode, hipStream_t st) {
    if (!wm.rows && wm.stride == kF && matrix_mode != VAD_MATRIX_F32) {
        using C = SplitCfg<LSTM>;
        int rc = ensure_rnn_attrs();
        if (rc != VAD_OK) return rc;
        const size_t lds_bytes = (size_t)C::lds_floats * sizeof(float);
        const long blocks = (n_windows + C::windows - 1) / C::windows;
        VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
        dispatch_split_mode(matrix_mode, [&](auto terms, auto f16) {
            hipLaunchKernelGGL((rnn_stream_split_kernel<LSTM, LARGE, decltype(terms)::value, decltype(f16)::value>),
                               dim3((unsigned)blocks), dim3(C::waves * 64), lds_bytes, st, x, n_windows, blob, p, dec, mode,
                               threshold);
        });
        VAD_HIP_CHECK(hipGetLastError());
        return VAD_OK;
    }
    if (!wm.rows && wm.stride == kF) {
        using C = StreamCfg<LSTM>;
        int rc = ensure_rnn_attrs();
        if (rc != VAD_OK) return rc;
        const size_t lds_bytes = (size_t)C::lds_floats * sizeof(float);
        const long blocks = (n_windows + C::windows - 1) / C::windows;
        VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
        hipLaunchKernelGGL((rnn_stream_kernel<LSTM, LARGE, KS>), dim3((unsigned)blocks), dim3(C::waves * 64), lds_bytes, st, x,
                           n_windows, blob, p, dec, mode, threshold);
        VAD_HIP_CHECK(hipGetLastError());
        return VAD_OK;
    }
    const NetLayout L = net_layout(LSTM, LARGE);
    const size_t lds_bytes = (size_t)L.head * sizeof(float);
    const long per_block = (long)kWindowsPerWave * kWavesPerBlock;
    const long blocks = (n_windows + per_block - 1) / per_block;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
    hipLaunchKernelGGL((rnn_kernel<LSTM, LARGE, KS>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds_bytes, st, x,
                       n_windows, wm, blob, p, dec, mode, threshold);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

template <bool LSTM, bool LARGE>
int launch(const float* x, long n_windows, WindowMap wm, const float* blob, int hidden, float* p, uint8_t* dec, int mode,
           float threshold, int matrix_mode, hipStream_t st) {
    return hidden_k_steps(hidden) <= kHk - 1
               ? launch_ks<LSTM, LARGE, kHk - 1>(x, n_windows, wm, blob, p, dec, mode, threshold, matrix_mode, st)
               : launch_ks<LSTM, LARGE, kHk>(x, n_windows, wm, blob, p, dec, mode, threshold, matrix_mode, st);
}

}  // namespace

static bool valid_matrix_mode(int m) {
    return m == VAD_MATRIX_F32 || m == VAD_MATRIX_BF16X6 || m == VAD_MATRIX_BF16X9 || m == VAD_MATRIX_BF16X8 ||
           m == VAD_MATRIX_F16X6 || m == VAD_MATRIX_F16X4;
}

int rnn_forward(bool lstm, const float* x, int64_t n_windows, WindowMap wm, const float* blob, int hidden, int large,
                float* p, uint8_t* dec, int mode, float threshold, void* stream, int matrix_mode = VAD_MATRIX_DEFAULT,
                float* ws_ptr = nullptr, size_t ws_bytes = 0) {
    const long wstride = wm.stride;
    VAD_REQUIRE(x && blob && p, "null pointer argument");
    VAD_REQUIRE(hidden >= 1 && hidden <= (lstm ? kHp : 3 * kHp),
                "hidden must be in 1..32 (LSTM) or 1..96 (GRU), got %d", hidden);
    VAD_REQUIRE(n_windows >= 0, "n_windows must be >= 0");
    VAD_REQUIRE(wstride >= 0 && (wstride % 4) == 0, "window_stride must be a non-negative multiple of 4 floats");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "x_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(blob) & 15) == 0, "blob_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(p) & 7) == 0, "p_dev must be 8-byte aligned");
    VAD_REQUIRE(mode == VAD_DECIDE_ARGMAX || mode == VAD_DECIDE_THRESHOLD, "unknown decision_mode %d", mode);
    VAD_REQUIRE(matrix_mode == VAD_MATRIX_DEFAULT || valid_matrix_mode(matrix_mode), "unknown matrix mode %d", matrix_mode);
    if (matrix_mode == VAD_MATRIX_DEFAULT) matrix_mode = g_matrix_mode.load(std::memory_order_relaxed);
    if (n_windows == 0) return VAD_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (lstm) {
        return large ? launch<true, true>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, matrix_mode, st)
                     : launch<true, false>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, matrix_mode, st);
    }
    const Workspace ws{ws_ptr, ws_bytes};
    if (hidden > 2 * kHp)
        return large ? launch_wide_kl<true, 3>(hidden, x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st)
                     : launch_wide_kl<false, 3>(hidden, x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st);
    if (hidden > kHp)
        return large ? launch_wide_kl<true, 2>(hidden, x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st)
                     : launch_wide_kl<false, 2>(hidden, x, n_windows, wm, blob, p, dec, mode, threshold, ws, matrix_mode, st);
    return large ? launch<false, true>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, matrix_mode, st)
                 : launch<false, false>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, matrix_mode, st);
}

}  // namespace vad

namespace vad {
namespace {
}  // namespace
}  // namespace vad

extern "C" int vad_set_matrix_mode(int mode) {
    VAD_REQUIRE(vad::valid_matrix_mode(mode), "unknown matrix mode %d", mode);
    vad::g_matrix_mode.store(mode, std::memory_order_relaxed);
    return VAD_OK;
}

extern "C" int vad_get_matrix_mode(void) { return vad::g_matrix_mode.load(std::memory_order_relaxed); }

#if defined(VAD_TIMELINE)
extern "C" int vad_debug_timeline(unsigned long long* out_host) {
    VAD_HIP_CHECK(hipDeviceSynchronize());
    VAD_HIP_CHECK(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vad::g_timeline), sizeof(vad::g_timeline)));
    return 2 * vad::rnn::kT * vad::kTlStamps;
}
#endif

extern "C" size_t vad_net_blob_floats(int lstm, int large, int hidden) {
    if (hidden < 1 || hidden > (lstm ? vad::kHp : 3 * vad::kHp)) return 0;
    return (size_t)vad::net_layout(lstm != 0, large != 0, vad::unit_tiles(hidden)).total;
}

extern "C" int vad_pack_net_weights(int lstm, int large, int hidden, const float* w_ih, const float* w_hh,
                                    const float* b_ih, const float* b_hh, const float* lin1_w, const float* lin1_b,
                                    const float* lin2_w, const float* lin2_b, const float* lin_w, const float* lin_b,
                                    float* blob, size_t blob_floats) {
    using namespace vad;
    VAD_REQUIRE(w_ih && w_hh && b_ih && b_hh && blob, "null pointer argument");
    VAD_REQUIRE(hidden >= 1 && hidden <= (lstm ? kHp : 3 * kHp),
                "hidden must be in 1..32 (LSTM) or 1..96 (GRU), got %d", hidden);
    if (large) VAD_REQUIRE(lin1_w && lin1_b && lin2_w && lin2_b, "large head needs lin1 and lin2");
    else VAD_REQUIRE(lin_w && lin_b, "small head needs lin");
    const int UT = unit_tiles(hidden);
    const NetLayout L = net_layout(lstm != 0, large != 0, UT);
    VAD_REQUIRE(blob_floats >= (size_t)L.total, "blob too small: need %d floats", L.total);
    const int H = hidden, G = L.gates, KH = UT * kHk;
    memset(blob, 0, sizeof(float) * (size_t)L.total);
    // tile row -> hidden unit (-1 = padding).  One-tile models may skip the all-padding k-step.
    auto unit_at = [&](int tile, int r) {
        const int u = unit_of_tile_row(r, tile_units(H, tile));
        return u < 0 ? -1 : 32 * tile + u;
    };
    // gate scale: sigmoid gates -log2 e, tanh gates 2 log2 e (GRU r,z | n ; LSTM i,f,o | g)
    auto gate_scale = [&](int g) { return (lstm ? g == 2 : g == 2) ? kTanhScale : kSigmoidScale; };
    for (int g = 0; g < G; ++g)
        for (int mt = 0; mt < UT; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = unit_at(mt, lane & 31), hh = lane >> 5;
                if (m < 0) continue;
                const double sc = gate_scale(g);
                for (int s = 0; s < kKs; ++s)
                    blob[L.a_ih + ((g * UT + mt) * kKs + s) * 64 + lane] =
                        (float)(sc * (double)w_ih[(g * H + m) * kF + s + 12 * hh]);
                for (int ks = 0; ks < KH; ++ks) {
                    const int u = unit_at(ks / kHk, mfma32_row(ks % kHk, hh));
                    if (u < 0) continue;
                    // one-tile models: k-step major; wider ones: groups of four k-steps per lane (WideA in rnn.hip)
                    const int at = UT == 1 ? ((g * UT + mt) * KH + ks) * 64 + lane
                                           : (((g * UT + mt) * KH + (ks & ~3)) * 64) + lane * 4 + (ks & 3);
                    blob[L.a_hh + at] = (float)(sc * (double)w_hh[(g * H + m) * H + u]);
                }
            }
    for (int mt = 0; mt < UT; ++mt)
        for (int hh = 0; hh < 2; ++hh)
            for (int i = 0; i < 16; ++i) {
                const int u = unit_at(mt, mfma32_row(i, hh));
                if (u < 0) continue;
                float* cb = blob + L.cb;
                auto at = [&](int a) -> float& { return cb[((a * UT + mt) * 2 + hh) * 16 + i]; };
                if (lstm) {
                    for (int g = 0; g < 4; ++g) at(g) = (float)(gate_scale(g) * ((double)b_ih[g * H + u] + (double)b_hh[g * H + u]));
                } else {
                    at(0) = (float)(kSigmoidScale * ((double)b_ih[u] + (double)b_hh[u]));                  // r
                    at(1) = (float)(kSigmoidScale * ((double)b_ih[H + u] + (double)b_hh[H + u]));          // z
                    at(2) = (float)(kTanhScale * (double)b_ih[2 * H + u]);                                 // W_in x + b_in
                    at(3) = (float)(kTanhScale * (double)b_hh[2 * H + u]);                                 // W_hn h + b_hn
                }
            }
    // head: rows of lin1 (26) or lin (2) as one 32-row A tile per k-step
    const float* hw = large ? lin1_w : lin_w;
    const int rows = large ? kL1 : 2;
    float* head = blob + L.head;
    for (int t = 0; t < kT; ++t)
        for (int ks = 0; ks < KH; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = lane & 31, u = unit_at(ks / kHk, mfma32_row(ks % kHk, lane >> 5));
                if (o < rows && u >= 0) head[(t * KH + ks) * 64 + lane] = hw[o * (kT * H) + t * H + u];
            }
    float* tail = blob + L.tail;
    if (large) {
        for (int hh = 0; hh < 2; ++hh)
            for (int i = 0; i < 16; ++i) {
                const int o = mfma32_row(i, hh);
                if (o >= kL1) continue;
                tail[hh * 16 + i] = lin1_b[o];
                tail[32 + (0 * 2 + hh) * 16 + i] = lin2_w[0 * kL1 + o];
                tail[32 + (1 * 2 + hh) * 16 + i] = lin2_w[1 * kL1 + o];
            }
        tail[96] = lin2_b[0];
        tail[97] = lin2_b[1];
    } else {
        tail[0] = lin_b[0];
        tail[1] = lin_b[1];
    }
    // float16 tables: weights times the largest power of two that keeps every one of them under float16's 65504
    // (rnn_device.h).  Sw may be negative for outsized weights; 13 already places |w| >= 2^-14 on the exact side.
    double wmax = 0.0;
    for (int g = 0; g < G; ++g)
        for (int i = 0; i < H * H; ++i) wmax = fmax(wmax, fabs(gate_scale(g) * (double)w_hh[(size_t)g * H * H + i]));
    for (int i = 0; i < rows * kT * H; ++i) wmax = fmax(wmax, fabs((double)hw[i]));
    int sw = 13;
    while (sw > -100 && ldexp(wmax, sw) > 65000.0) --sw;
    VAD_REQUIRE(wmax - wmax == 0.0 && sw > -100, "weights are not finite");
    const float wscale = ldexpf(1.0f, sw);
    tail[kTailScale(large != 0) + 0] = ldexpf(1.0f, kStateShift + sw);
    tail[kTailScale(large != 0) + 1] = ldexpf(1.0f, -(kStateShift + sw));
    tail[kTailScale(large != 0) + 2] = wscale;
    bool f16_finite = true;
    for (int fmt = 0; fmt < 2; ++fmt) {
        const bool f16 = fmt == 1;
        const float ws = f16 ? wscale : 1.0f;
        const int off_hh = f16 ? L.h_hh : L.s_hh, off_head = f16 ? L.h_head : L.s_head;
        if (UT == 1) {
            // (hi, mid, lo) images of the float32 A operands above as v_mfma_f32_16x16x32_{bf16,f16} A operands: sub-tile
            // mi, lane l = 16 q + m holds tile row 16 mi + m, k-slots 8 q + j = tile rows split_row_of_k_slot(q, j); in the
            // float32 image tile row p is k-step (p & 3) + 4 (p >> 3) of lane half (p >> 2) & 1
            auto split_tile = [&](const float* a16, float* dst) {      // a16: [16][64] floats -> [2][3][64][8 x 16 bit]
                uint16_t* out = reinterpret_cast<uint16_t*>(dst);
                for (int mi = 0; mi < 2; ++mi)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int m = 16 * mi + (lane & 15), p = split_row_of_k_slot(lane >> 4, j);
                            uint16_t t3[3];
                            split3_host(ws * a16[((p & 3) + 4 * (p >> 3)) * 64 + m + 32 * ((p >> 2) & 1)], f16, t3, &f16_finite);
                            for (int term = 0; term < 3; ++term) out[(((mi * 3 + term) * 64 + lane) * 8) + j] = t3[term];
                        }
            };
            for (int g = 0; g < G; ++g) split_tile(blob + L.a_hh + g * kHk * 64, blob + off_hh + g * kSplitTile);
            for (int t = 0; t < kT; ++t) split_tile(blob + L.head + t * kHk * 64, blob + off_head + t * kSplitTile);
        } else {
            // wide GRU.  Entry (rows 16 mi .. + 15 of unit tile mt, k-block kb): lane l = 16 q + m holds tile row 16 mi + m
            // of tile mt and, as k-slots 8 q + j, the tile rows split_row_of_k_slot(q, j) of tile kb (rnn_device.h); rows
            // and slots that carry no unit stay zero.
            auto entry = [&](int off, int index) { return reinterpret_cast<uint16_t*>(blob + off + (size_t)index * kSplitEntry); };
            auto put = [&](float x, uint16_t* e_hi, uint16_t* e_mid, uint16_t* e_lo, int at) {
                uint16_t t3[3];
                split3_host(ws * x, f16, t3, &f16_finite);
                e_hi[at] = t3[0];
                e_mid[at] = t3[1];
                e_lo[at] = t3[2];
            };
            for (int g = 0; g < 3; ++g)
                for (int mt = 0; mt < UT; ++mt)
                    for (int mi = 0; mi < 2; ++mi)
                        for (int kb = 0; kb < UT; ++kb) {
                            const int base = wide_split_base(UT, g, mt, mi, kb);
                            uint16_t* e_hi = entry(off_hh, base * 2 + 0);
                            uint16_t* e_mid = entry(off_hh, base * 2 + 1);
                            uint16_t* e_lo = entry(off_hh, wide_split_hm_entries(UT) + base);
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int m = unit_at(mt, 16 * mi + (lane & 15));
                                    const int u = unit_at(kb, split_row_of_k_slot(lane >> 4, j));
                                    const float x = (m < 0 || u < 0) ? 0.0f : (float)(gate_scale(g) * (double)w_hh[(g * H + m) * H + u]);
                                    put(x, e_hi, e_mid, e_lo, lane * 8 + j);
                                }
                        }
            for (int t = 0; t < kT; ++t)
                for (int mi = 0; mi < 2; ++mi)
                    for (int kb = 0; kb < UT; ++kb) {
                        const int base = ((t * 2 + mi) * UT + kb) * 3;
                        uint16_t* e_hi = entry(off_head, base + 0);
                        uint16_t* e_mid = entry(off_head, base + 1);
                        uint16_t* e_lo = entry(off_head, base + 2);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int o = 16 * mi + (lane & 15);
                                const int u = unit_at(kb, split_row_of_k_slot(lane >> 4, j));
                                const float x = (o < rows && u >= 0) ? hw[o * (kT * H) + t * H + u] : 0.0f;
                                put(x, e_hi, e_mid, e_lo, lane * 8 + j);
                            }
                    }
        }
    }
    VAD_REQUIRE(f16_finite, "a weight does not fit the float16 tables");     // cannot happen: wscale was chosen for it
    return VAD_OK;
}

extern "C" int vad_gru_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                   int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                   float threshold, void* stream) {
    return vad::rnn_forward(false, x_dev, n_windows, vad::WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden,
                            large, p_dev, decision_dev, decision_mode, threshold, stream);
}

extern "C" size_t vad_gru_workspace_bytes(int hidden, int64_t n_windows) {
    if (hidden <= vad::kHp || hidden > 3 * vad::kHp || n_windows <= 0) return 0;
    return (size_t)(n_windows + vad::kT - 1) * 3 * vad::unit_tiles(hidden) * 32 * sizeof(float);
}

extern "C" int vad_gru_forward_ws_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                      int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                      float threshold, void* workspace_dev, size_t workspace_bytes, void* stream) {
    using namespace vad;
    VAD_REQUIRE(workspace_dev || workspace_bytes == 0, "null workspace with a non-zero size");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(workspace_dev) & 15) == 0, "workspace_dev must be 16-byte aligned");
    return rnn_forward(false, x_dev, n_windows, WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden, large, p_dev,
                       decision_dev, decision_mode, threshold, stream, VAD_MATRIX_DEFAULT, static_cast<float*>(workspace_dev),
                       workspace_bytes);
}

extern "C" int vad_gru_forward_mode_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                        int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                        float threshold, int matrix_mode, void* workspace_dev, size_t workspace_bytes,
                                        void* stream) {
    using namespace vad;
    VAD_REQUIRE(workspace_dev || workspace_bytes == 0, "null workspace with a non-zero size");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(workspace_dev) & 15) == 0, "workspace_dev must be 16-byte aligned");
    return rnn_forward(false, x_dev, n_windows, WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden, large, p_dev,
                       decision_dev, decision_mode, threshold, stream, matrix_mode, static_cast<float*>(workspace_dev),
                       workspace_bytes);
}

extern "C" int vad_lstm_forward_mode_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                         int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                         float threshold, int matrix_mode, void* stream) {
    return vad::rnn_forward(true, x_dev, n_windows, vad::WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden,
                            large, p_dev, decision_dev, decision_mode, threshold, stream, matrix_mode);
}

extern "C" int vad_lstm_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                    int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                    float threshold, void* stream) {
    return vad::rnn_forward(true, x_dev, n_windows, vad::WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden,
                            large, p_dev, decision_dev, decision_mode, threshold, stream);
}

namespace vad {
int window_map_for_rows(int64_t n_rows, const int64_t* window_rows_dev, int64_t n_windows, WindowMap* wm) {
    VAD_REQUIRE(n_rows >= VAD_FRAMES, "the feature matrix needs at least %d rows, got %ld", VAD_FRAMES, (long)n_rows);
    VAD_REQUIRE(window_rows_dev || n_windows == 0, "null pointer argument");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(window_rows_dev) & 7) == 0, "window_rows_dev must be 8-byte aligned");
    *wm = WindowMap{VAD_FEATURES, window_rows_dev, (long)n_rows - VAD_FRAMES};
    return VAD_OK;
}
}  // namespace vad

extern "C" int vad_gru_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                        int64_t n_windows, const float* blob_dev, int hidden, int large, float* p_dev,
                                        uint8_t* decision_dev, int decision_mode, float threshold, void* stream) {
    vad::WindowMap wm;
    const int rc = vad::window_map_for_rows(n_rows, window_rows_dev, n_windows, &wm);
    if (rc != VAD_OK) return rc;
    return vad::rnn_forward(false, feat_dev, n_windows, wm, blob_dev, hidden, large, p_dev, decision_dev, decision_mode,
                            threshold, stream);
}

extern "C" int vad_lstm_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                         int64_t n_windows, const float* blob_dev, int hidden, int large, float* p_dev,
                                         uint8_t* decision_dev, int decision_mode, float threshold, void* stream) {
    vad::WindowMap wm;
    const int rc = vad::window_map_for_rows(n_rows, window_rows_dev, n_windows, &wm);
    if (rc != VAD_OK) return rc;
    return vad::rnn_forward(true, feat_dev, n_windows, wm, blob_dev, hidden, large, p_dev, decision_dev, decision_mode,
                            threshold, stream);
}
