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

#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "rnn_device.h"

namespace vad {

using namespace rnn;

// How the stride-1 GRU / LSTM kernels form their float32 matrix products (vad_set_matrix_mode).
std::atomic<int> g_matrix_mode{VAD_MATRIX_F32};

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

template <bool LSTM>
struct SplitCfg : StreamGeom<LSTM, LSTM ? 7 : 8> {
    using B = StreamGeom<LSTM, LSTM ? 7 : 8>;
    static constexpr int lds_floats = B::gi_floats + B::G * kSplitTile + 4 * 2 * 16;
};

template <bool LSTM, bool LARGE, int TERMS>
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
    for (int i = threadIdx.x; i < G * kSplitTile; i += C::waves * 64) s_hh[i] = blob[L.s_hh + i];
    for (int i = threadIdx.x; i < 4 * 2 * 16; i += C::waves * 64) cb[i] = blob[L.cb + i];

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wave = threadIdx.x >> 6;
    const long w0 = (long)blockIdx.x * C::windows;
    const long last_row = n_windows + kT - 2;

    // ---- phase 1: gi[row][gate * 32 + unit] = bias + W_ih x_row (float32 MFMA, as rnn_stream_kernel) ----
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

    // ---- phase 2: 30 steps for the wave's 32 windows.  A wave alternates between a matrix phase (18 or 27
    // bf16 MFMAs per tile) and a vector phase (gate math, bf16 split); the two waves that share a SIMD
    // (w and w + 4) overlap only while they are in different phases, and left alone they start in lockstep,
    // so the second one is delayed once by about half a step. ----
    const long w = w0 + wave * kWindowsPerWave + col;
    if (w0 + wave * kWindowsPerWave >= n_windows) return;       // wave-uniform, after the only barrier
    const float* a1 = blob + L.head;
    const u32x4* s_gate = reinterpret_cast<const u32x4*>(s_hh);
    const u32x4* s_head = reinterpret_cast<const u32x4*>(blob + L.s_head);
    f32x16 h, c, lacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { h[i] = 0.0f; c[i] = 0.0f; lacc[i] = 0.0f; }
    const float* girow = gi + (wave * kWindowsPerWave + col) * C::rs + 4 * half;
    SplitB b;
#ifndef VAD_SPLIT_STAGGER
#define VAD_SPLIT_STAGGER 14
#endif
    if ((wave >> 2) != 0 && VAD_SPLIT_STAGGER > 0) __builtin_amdgcn_s_sleep(VAD_SPLIT_STAGGER);   // x 64 cycles

    for (int t = 0; t < kT; ++t) {
        HeadWeights<false, kHk> hw;                              // 2-row head: vector ALU, float32 weights
        if (!LARGE) head_load<false, kHk>(hw, a1, t, lane);
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
        if (t > 0) {                                             // b holds the split of h_{t-1}
            if (LARGE) lacc = split_tile_mfma<TERMS>(lacc, s_head + (t - 1) * (kSplitTile / 4), b, lane);
            if (LSTM) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = split_tile_mfma<TERMS>(acc[g], s_gate + g * (kSplitTile / 4), b, lane);
            } else {
                acc[0] = split_tile_mfma<TERMS>(acc[0], s_gate + 0 * (kSplitTile / 4), b, lane);
                acc[1] = split_tile_mfma<TERMS>(acc[1], s_gate + 1 * (kSplitTile / 4), b, lane);
                acc[3] = split_tile_mfma<TERMS>(acc[3], s_gate + 2 * (kSplitTile / 4), b, lane);
            }
        }
        gate_update<LSTM>(acc, h, c);
        split_state(b, h);                                       // for this step's head and the next step's gates
        if (!LARGE) head_step<false, kHk>(lacc, hw, h);
    }
    if (LARGE) lacc = split_tile_mfma<TERMS>(lacc, s_head + (kT - 1) * (kSplitTile / 4), b, lane);
    head_finish<LARGE>(lacc);
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

// ---------------------------------------------------------------------------------------------
// GRU with 33..96 hidden units (UT = 2 or 3 unit tiles), any window stride.  Same cell as rnn_kernel,
// one 32-unit tile of the new state at a time: the accumulators of a tile (r, z, n-input, n-hidden)
// see the whole previous state as B operand, so the new state is double-buffered in registers.

template <bool LARGE, int UT>
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
            if (t > 0) {
#pragma unroll
                for (int kt = 0; kt < UT; ++kt) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < kHk; ++i) {
                        const float hv = h[kt][i];
                        const int ks = kt * kHk + i;
                        acc[0] = MFMA(a_hh[((0 * UT + mt) * UT * kHk + ks) * 64 + lane], hv, acc[0]);
                        acc[1] = MFMA(a_hh[((1 * UT + mt) * UT * kHk + ks) * 64 + lane], hv, acc[1]);
                        acc[3] = MFMA(a_hh[((2 * UT + mt) * UT * kHk + ks) * 64 + lane], hv, acc[3]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float r = sigmoid_pre(acc[0][i]);
                const float z = sigmoid_pre(acc[1][i]);
                const float n = tanh_pre(fmaf(r, acc[3][i], acc[2][i]));
                hn[mt][i] = fmaf(z, h[mt][i] - n, n);
            }
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) h[u] = hn[u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < UT; ++kt)
#pragma unroll
            for (int i = 0; i < kHk; ++i) lacc = MFMA(a1[(t * UT * kHk + kt * kHk + i) * 64 + lane], h[kt][i], lacc);
    }
    head_epilogue<LARGE>(lacc, blob + L.tail, half, w, n_windows, p_out, dec_out, mode, threshold);
}

template <typename K>
int allow_lds(K kernel, int floats) {
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      floats * (int)sizeof(float)));
    return VAD_OK;
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
    if ((rc = allow_lds(&rnn_stream_split_kernel<false, false, 6>, SplitCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<false, true, 6>, SplitCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<true, false, 6>, SplitCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<true, true, 6>, SplitCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<false, false, 9>, SplitCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<false, true, 9>, SplitCfg<false>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<true, false, 9>, SplitCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&rnn_stream_split_kernel<true, true, 9>, SplitCfg<true>::lds_floats))) return rc;
    if ((rc = allow_lds(&gru_wide_kernel<false, 2>, net_layout(false, false, 2).head))) return rc;
    if ((rc = allow_lds(&gru_wide_kernel<true, 2>, net_layout(false, true, 2).head))) return rc;
    if ((rc = allow_lds(&gru_wide_kernel<false, 3>, net_layout(false, false, 3).head))) return rc;
    if ((rc = allow_lds(&gru_wide_kernel<true, 3>, net_layout(false, true, 3).head))) return rc;
    done[dev] = true;
    return VAD_OK;
}

namespace {

template <bool LARGE, int UT>
int launch_wide(const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec, int mode,
                float threshold, hipStream_t st) {
    const NetLayout L = net_layout(false, LARGE, UT);
    const size_t lds_bytes = (size_t)L.head * sizeof(float);
    int rc = ensure_rnn_attrs();
    if (rc != VAD_OK) return rc;
    const long per_block = (long)kWindowsPerWave * kWavesPerBlock;
    const long blocks = (n_windows + per_block - 1) / per_block;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
    hipLaunchKernelGGL((gru_wide_kernel<LARGE, UT>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds_bytes, st, x,
                       n_windows, wm, blob, p, dec, mode, threshold);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

// ---------------------------------------------------------------------------------------------
// launchers

template <bool LSTM, bool LARGE, int KS>
int launch_ks(const float* x, long n_windows, WindowMap wm, const float* blob, float* p, uint8_t* dec, int mode,
           float threshold, hipStream_t st) {
    const int matrix_mode = g_matrix_mode.load(std::memory_order_relaxed);
    if (!wm.rows && wm.stride == kF && matrix_mode != VAD_MATRIX_F32) {
        using C = SplitCfg<LSTM>;
        int rc = ensure_rnn_attrs();
        if (rc != VAD_OK) return rc;
        const size_t lds_bytes = (size_t)C::lds_floats * sizeof(float);
        const long blocks = (n_windows + C::windows - 1) / C::windows;
        VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many windows for one launch");
        if (matrix_mode == VAD_MATRIX_BF16X9)
            hipLaunchKernelGGL((rnn_stream_split_kernel<LSTM, LARGE, 9>), dim3((unsigned)blocks), dim3(C::waves * 64),
                               lds_bytes, st, x, n_windows, blob, p, dec, mode, threshold);
        else
            hipLaunchKernelGGL((rnn_stream_split_kernel<LSTM, LARGE, 6>), dim3((unsigned)blocks), dim3(C::waves * 64),
                               lds_bytes, st, x, n_windows, blob, p, dec, mode, threshold);
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
           float threshold, hipStream_t st) {
    return hidden_k_steps(hidden) == kHk - 1
               ? launch_ks<LSTM, LARGE, kHk - 1>(x, n_windows, wm, blob, p, dec, mode, threshold, st)
               : launch_ks<LSTM, LARGE, kHk>(x, n_windows, wm, blob, p, dec, mode, threshold, st);
}

}  // namespace

int rnn_forward(bool lstm, const float* x, int64_t n_windows, WindowMap wm, const float* blob, int hidden, int large,
                float* p, uint8_t* dec, int mode, float threshold, void* stream) {
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
    if (n_windows == 0) return VAD_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (lstm) {
        return large ? launch<true, true>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, st)
                     : launch<true, false>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, st);
    }
    if (hidden > 2 * kHp)
        return large ? launch_wide<true, 3>(x, n_windows, wm, blob, p, dec, mode, threshold, st)
                     : launch_wide<false, 3>(x, n_windows, wm, blob, p, dec, mode, threshold, st);
    if (hidden > kHp)
        return large ? launch_wide<true, 2>(x, n_windows, wm, blob, p, dec, mode, threshold, st)
                     : launch_wide<false, 2>(x, n_windows, wm, blob, p, dec, mode, threshold, st);
    return large ? launch<false, true>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, st)
                 : launch<false, false>(x, n_windows, wm, blob, hidden, p, dec, mode, threshold, st);
}

}  // namespace vad

extern "C" int vad_set_matrix_mode(int mode) {
    VAD_REQUIRE(mode == VAD_MATRIX_F32 || mode == VAD_MATRIX_BF16X6 || mode == VAD_MATRIX_BF16X9,
                "unknown matrix mode %d", mode);
    vad::g_matrix_mode.store(mode, std::memory_order_relaxed);
    return VAD_OK;
}

extern "C" int vad_get_matrix_mode(void) { return vad::g_matrix_mode.load(std::memory_order_relaxed); }

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
    auto unit_at = [&](int tile, int r) { return UT == 1 ? unit_of_row(r, H) : (32 * tile + r < H ? 32 * tile + r : -1); };
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
                    if (u >= 0)
                        blob[L.a_hh + ((g * UT + mt) * KH + ks) * 64 + lane] = (float)(sc * (double)w_hh[(g * H + m) * H + u]);
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
    if (UT == 1) {
        // bf16 (hi, mid, lo) images of the float32 A operands above: element j of k-step s (of 16) is the
        // float32 k-step 8 s + j of the same lane
        auto split_tile = [&](const float* a16, float* dst) {      // a16: [16][64] floats -> [2][3][64][8 bf16]
            uint16_t* out = reinterpret_cast<uint16_t*>(dst);
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float x = a16[(8 * s + j) * 64 + lane];
                        uint32_t bits;
                        memcpy(&bits, &x, 4);
                        const uint32_t hb = bits & 0xFFFF0000u;
                        float hi;
                        memcpy(&hi, &hb, 4);
                        const float r1 = x - hi;
                        memcpy(&bits, &r1, 4);
                        const uint32_t mb = bits & 0xFFFF0000u;
                        float mid;
                        memcpy(&mid, &mb, 4);
                        const float r2 = r1 - mid;
                        memcpy(&bits, &r2, 4);
                        out[(((s * 3 + 0) * 64 + lane) * 8) + j] = (uint16_t)(hb >> 16);
                        out[(((s * 3 + 1) * 64 + lane) * 8) + j] = (uint16_t)(mb >> 16);
                        out[(((s * 3 + 2) * 64 + lane) * 8) + j] = (uint16_t)(bits >> 16);
                    }
        };
        for (int g = 0; g < G; ++g) split_tile(blob + L.a_hh + g * kHk * 64, blob + L.s_hh + g * kSplitTile);
        for (int t = 0; t < kT; ++t) split_tile(blob + L.head + t * kHk * 64, blob + L.s_head + t * kSplitTile);
    }
    return VAD_OK;
}

extern "C" int vad_gru_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                   int hidden, int large, float* p_dev, uint8_t* decision_dev, int decision_mode,
                                   float threshold, void* stream) {
    return vad::rnn_forward(false, x_dev, n_windows, vad::WindowMap{(long)window_stride, nullptr, 0}, blob_dev, hidden,
                            large, p_dev, decision_dev, decision_mode, threshold, stream);
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
