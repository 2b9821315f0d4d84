// Conv + GRU window classifier for gfx950: the reference's NickNet (core/models.py:245-330) in its default
// configuration -- gated convolutions (GatedConv, core/models.py:180-202), no residual connections, a
// uni-directional GRU (BiRNN with lstm=False, bidir=False, core/models.py:84-128) and the fully connected
// `embed` head -- in eval mode (BatchNorm running statistics folded into the weights, Dropout off).
//
//   large: Conv 24 -> 32 -> 28 -> 25 -> 18 (kernel 3, padding 1, tanh(conv) * sigmoid(conv_gate)),
//          GRU 18 -> 29 -> 29 (2 layers), embed 58 -> 32 -> 18 -> 2, softmax          29 605 parameters
//   small: Conv 24 -> 20 -> 18 -> 16, GRU 16 -> 21 (1 layer), embed 21 -> 18 -> 2      10 019 parameters
//
// The reference's own forward raises between the GRU and `embed` (BiRNN.forward flattens all 30 outputs
// where `embed` takes L * H columns, core/models.py:166-174).  Here the head input is the final hidden state
// of every GRU layer, h_n.permute(1, 0, 2).reshape(B, L * H): the one tensor of the GRU with that width.
//
// Mapping.  One wave owns 16 windows; every matrix product is computed transposed on the float32 matrix
// cores with v_mfma_f32_16x16x4_f32,
//
//     Out^T (16 units x 16 windows)  +=  W (16 units x 4 k)  .  In^T (4 k x 16 windows)
//
// Lane 16 q + n holds rows 4 q + i (i = 0..3) of the 16 x 16 result for window n, and as B operand the same
// lane supplies k = q.  A layer output ("tile": up to 32 units in two 16-row blocks, 8 registers per lane)
// whose unit 16 b + 4 i + q sits on row 4 q + i of block b is therefore at once the B operand of the next
// layer: k-step (b, i) takes register [b][i] of every lane, and the weight tables carry their columns in that
// order.  Nothing leaves the registers between the layers: the three convolution taps read the tiles of
// the previous layer at positions p - 1, p, p + 1, which the wave keeps in a three-deep register ring, and
// the layers run as a pipeline over the position -- conv1 at p, conv2 at p - 1, ... , the GRU step at
// p - (layers - 1) -- so no (B, C, 30) activation ever exists.  Weights are A operands, one 16-byte entry of
// four k-steps per lane, in a block-shared LDS copy (147 KB for the large model: one block per CU, whose
// waves stride over the window tiles of the launch).
//
// float32 MFMA is a k-ordered FMA chain.  The products whose B operand is a hidden state -- the GRU's recurrent
// products and the input products of its second layer -- run on the 16-bit matrix pipe instead, each float32
// product as six float16 partial products of exactly split operands with float32 accumulation (what is left out is
// below 2^-35 of a product; nicknet_device.h, rnn_device.h): 16 cycles per K = 32 instead of 8 x 32.  tanh / sigmoid
// are v_exp_f32 / v_rcp_f32 forms on pre-scaled pre-activations (the packer multiplies the sigmoid inputs' weights
// by -log2 e and the tanh inputs' by 2 log2 e), as in the GRU / LSTM kernels.


#include <mutex>

#include "nicknet_device.h"

namespace vad {
namespace nn {

// Optional outputs of the sub-modules, in the reference's layouts (diagnostics; any may be null).
struct StageOut {
    float* trunk;      // (B, C, 30)   conv trunk, core/models.py:308-323
    float* seq;        // (B, 30, H)   outputs of the last GRU layer, core/models.py:162
    float* hn;         // (L, B, H)    final hidden state of every GRU layer
    float* logits;     // (B, 2)       embed output before the softmax
};

template <bool LARGE>
__global__ __launch_bounds__(kWaves * 64) void nicknet_kernel(const float* __restrict__ x, long n_windows, WindowMap wm,
                                                              const float* __restrict__ trunk_in,
                                                              const float* __restrict__ blob, float* __restrict__ p_out,
                                                              uint8_t* __restrict__ dec_out, int mode, float threshold,
                                                              StageOut so) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    constexpr int NC = D::nc, C = D::chan(NC), H = D::hidden;
    extern __shared__ __align__(16) float lds[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(blob);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < LY.lds / 4; i += kWaves * 64) dst[i] = src[i];
    }
    __syncthreads();                                             // the only barrier: waves are independent from here on

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, n = lane & 15;
    const f32x4* a_lane = reinterpret_cast<const f32x4*>(lds) + lane;
    const long n_tiles = (n_windows + kWaveWindows - 1) / kWaveWindows;

    for (long tile = (long)blockIdx.x * kWaves + (threadIdx.x >> 6); tile < n_tiles; tile += (long)gridDim.x * kWaves) {
        const long w = tile * kWaveWindows + n;
        const long wc = w < n_windows ? w : n_windows - 1;       // clamp the loads, mask the stores
        const float* xw = trunk_in ? nullptr : window_ptr(x, wm, wc) + kXk * q;

        float xp[kXk], xc[kXk], xn[kXk];
#pragma unroll
        for (int j = 0; j < kXk; ++j) xp[j] = xc[j] = xn[j] = 0.0f;
        if (!trunk_in) {
            load_x_row(xc, xw);
            load_x_row(xn, xw + kF);
        }
        Tile ring[NC - 1][3];
#pragma unroll
        for (int l = 0; l < NC - 1; ++l)
#pragma unroll
            for (int s = 0; s < 3; ++s) ring[l][s].b[0] = ring[l][s].b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        Tile h[D::layers];
        SplitTile hs[D::layers];                                 // written by every step before it is read
#pragma unroll
        for (int l = 0; l < D::layers; ++l) h[l].b[0] = h[l].b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < D::layers; ++l) hs[l].hi = hs[l].mid = hs[l].lo = rnn::u32x4{0u, 0u, 0u, 0u};

#pragma unroll 1
        for (int p = 0; p < kT + NC - 1; ++p) {
            const int t = p - (NC - 1);                          // position the last conv layer and the GRU are at
            Tile cur;
            cur.b[0] = cur.b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (!trunk_in) {
                float xl[kXk];                                   // row p + 2, used from the next iteration on
#pragma unroll
                for (int j = 0; j < kXk; ++j) xl[j] = 0.0f;
                if (p + 2 < kT) load_x_row(xl, xw + (p + 2) * kF);
                if (p < kT) cur = conv_layer<LARGE, 0>(a_lane, lds, xp, xc, xn, p >= 1, p + 1 < kT, q);
#pragma unroll
                for (int j = 0; j < kXk; ++j) {
                    xp[j] = xc[j];
                    xc[j] = xn[j];
                    xn[j] = xl[j];
                }
                // conv layer l is at position p - l and reads layer l - 1 at p - l - 1, p - l, p - l + 1: the ring of
                // layer l - 1 after this iteration's shift
                NN_STAGE_FENCE;
                conv_stage<LARGE, 1>(ring[0], cur, p, a_lane, lds, q);
                NN_STAGE_FENCE;
                conv_stage<LARGE, 2>(ring[1], cur, p, a_lane, lds, q);
                NN_STAGE_FENCE;
                if constexpr (NC > 3) conv_stage<LARGE, 3>(ring[2], cur, p, a_lane, lds, q);
                NN_STAGE_FENCE;
            } else if (t >= 0) {                                 // the trunk's output is given: (B, C, 30)
#pragma unroll
                for (int blk = 0; blk < n_blocks(C); ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 16 * blk + 4 * i + q;
                        cur.b[blk][i] = c < C ? trunk_in[(wc * C + c) * kT + t] : 0.0f;
                    }
            }
            if (t < 0) continue;
            if (so.trunk && w < n_windows) {
#pragma unroll
                for (int blk = 0; blk < n_blocks(C); ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 16 * blk + 4 * i + q;
                        if (c < C) so.trunk[(w * C + c) * kT + t] = cur.b[blk][i];
                    }
            }
            NN_STAGE_FENCE;
            gru_layers<LARGE>(h, hs, cur, a_lane, lds, t == 0, q);
            if (so.seq && w < n_windows) {
#pragma unroll
                for (int blk = 0; blk < n_blocks(H); ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = 16 * blk + 4 * i + q;
                        if (u < H) so.seq[(w * kT + t) * H + u] = h[D::layers - 1].b[blk][i];
                    }
            }
        }
        if (so.hn && w < n_windows) {
#pragma unroll
            for (int l = 0; l < D::layers; ++l)
#pragma unroll
                for (int blk = 0; blk < n_blocks(H); ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = 16 * blk + 4 * i + q;
                        if (u < H) so.hn[((long)l * n_windows + w) * H + u] = h[l].b[blk][i];
                    }
        }
        embed_head<LARGE>(h, blob, lane, w, n_windows, p_out, dec_out, mode, threshold, so.logits);
    }
}

// embed alone: rows of v (n, L * H) -> logits (and posteriors), for checking the head against the reference's
// net.rnn.embed on its own.
template <bool LARGE>
__global__ __launch_bounds__(256) void nicknet_embed_kernel(const float* __restrict__ v, long n_rows, const float* __restrict__ blob,
                                                            float* __restrict__ logits_out, float* __restrict__ p_out) {
    using D = Dims<LARGE>;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, n = lane & 15;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * kWaveWindows >= n_rows) return;
    const long w = tile * kWaveWindows + n;
    const long wc = w < n_rows ? w : n_rows - 1;
    Tile h[D::layers];
#pragma unroll
    for (int l = 0; l < D::layers; ++l)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 16 * blk + 4 * i + q;
                h[l].b[blk][i] = u < D::hidden ? v[wc * D::head_cols + l * D::hidden + u] : 0.0f;
            }
    embed_head<LARGE>(h, blob, lane, w, n_rows, p_out, nullptr, VAD_DECIDE_ARGMAX, 0.0f, logits_out);
}

// ---------------------------------------------------------------------------------------------
// host: BatchNorm folding and packing

constexpr double kBnEps = 1e-5;

struct Affine {          // Conv1d / Linear (weight, bias) followed by BatchNorm1d (eval): y = s (W x + b - mean) + beta
    const float *w, *b, *gamma, *beta, *mean, *var;
    double scale(int o) const { return (double)gamma[o] / sqrt((double)var[o] + kBnEps); }
    double shift(int o) const { return ((double)b[o] - (double)mean[o]) * scale(o) + (double)beta[o]; }
};

// unit held by (k-step ks, lane group q) of a tile with nu units, or -1
inline int tile_unit(int nu, int ks, int q) {
    const int k0 = block_ksteps(nu, 0);
    const int b = ks < k0 ? 0 : 1, i = b ? ks - k0 : ks;
    const int u = 16 * b + 4 * i + q;
    return (i < block_ksteps(nu, b) && u < nu) ? u : -1;
}
// unit on row m of out block blk, or -1
inline int out_unit(int nu, int blk, int m) {
    const int u = 16 * blk + 4 * (m & 3) + (m >> 2);
    return u < nu ? u : -1;
}

template <typename W>
void put_table(float* dst, int ks_total, int nout, W weight) {       // weight(out block, row m, k-step, lane group) -> double
    for (int g = 0; g < groups(ks_total); ++g)
        for (int o = 0; o < nout; ++o)
            for (int lane = 0; lane < 64; ++lane) {
                const int width = g < groups(ks_total) - 1 ? 4 : last_width(ks_total);
                for (int j = 0; j < width; ++j) {
                    const int ks = 4 * g + j;
                    dst[entry_offset(ks_total, nout, g, o, lane) + j] = ks < ks_total ? (float)weight(o, lane & 15, ks, lane >> 4) : 0.0f;
                }
            }
}

template <typename B>
void put_bias(float* dst, int nu, B bias) {                            // bias(unit) -> double
    for (int blk = 0; blk < 2; ++blk)
        for (int qq = 0; qq < 4; ++qq)
            for (int i = 0; i < 4; ++i) {
                const int u = 16 * blk + 4 * i + qq;
                dst[(blk * 4 + qq) * 4 + i] = u < nu ? (float)bias(u) : 0.0f;
            }
}

template <bool LARGE>
int pack(const float* const* t, float* blob) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    memset(blob, 0, sizeof(float) * (size_t)LY.total);
    int idx = 0;
    auto affine = [&]() {
        Affine a{t[idx], t[idx + 1], t[idx + 2], t[idx + 3], t[idx + 4], t[idx + 5]};
        idx += 6;
        return a;
    };
    for (int l = 0; l < D::nc; ++l) {
        const Affine val = affine(), gate = affine();            // conv (tanh), conv_gate (sigmoid): weight (cout, cin, 3)
        const int cin = D::chan(l), cout = D::chan(l + 1), nb = n_blocks(cout);
        for (int tap = 0; tap < 3; ++tap)
            put_table(blob + LY.conv[l] + tap * LY.conv_tap[l], D::conv_ks(l), 2 * nb, [&](int o, int m, int ks, int qq) {
                const Affine& a = o < nb ? val : gate;
                const int ou = out_unit(cout, o % nb, m);
                const int iu = l == 0 ? kXk * qq + ks : tile_unit(cin, ks, qq);
                if (ou < 0 || iu < 0) return 0.0;
                return (o < nb ? rnn::kTanhScale : rnn::kSigmoidScale) * a.scale(ou) * (double)a.w[(ou * cin + iu) * 3 + tap];
            });
        put_bias(blob + LY.conv_bias[l], cout, [&](int u) { return rnn::kTanhScale * val.shift(u); });
        put_bias(blob + LY.conv_bias[l] + kBiasTile, cout, [&](int u) { return rnn::kSigmoidScale * gate.shift(u); });
    }
    constexpr int H = D::hidden, NB = n_blocks(H);
    for (int g = 0; g < D::layers; ++g) {
        const float *w_ih = t[idx], *w_hh = t[idx + 1], *b_ih = t[idx + 2], *b_hh = t[idx + 3];
        idx += 4;
        const int in = D::gru_in(g);
        auto gate_scale = [](int gate) { return gate == 2 ? rnn::kTanhScale : rnn::kSigmoidScale; };
        // A weight matrix (3 H x cols) that multiplies a hidden state: float16 split A operands of
        // v_mfma_f32_16x16x32_f16.  Lane 16 qa + m holds row m of the out block and, as k-slots 8 qa + j, the units lane
        // group qa keeps in register [j >> 2][j & 3] of a tile; times the largest power of two that keeps every weight
        // under float16's 65504 (rnn_device.h).  scale[0] = 2^-(14 + Sw) undoes it and the state's 2^14.
        auto put_split = [&](const float* wm, int cols, float* dst, float* scale) {
            double wmax = 0.0;
            for (int gate = 0; gate < 3; ++gate)
                for (int i = 0; i < H * cols; ++i) wmax = fmax(wmax, fabs(gate_scale(gate) * (double)wm[gate * H * cols + i]));
            int sw = 13;
            while (sw > -100 && ldexp(wmax, sw) > 65000.0) --sw;
            if (!(wmax - wmax == 0.0) || sw <= -100) return false;
            uint16_t* out = reinterpret_cast<uint16_t*>(dst);
            bool finite = true;
            for (int o = 0; o < 3 * NB; ++o)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int gate = o / NB, ou = out_unit(H, o % NB, lane & 15);
                        const int iu = 16 * (j >> 2) + 4 * (j & 3) + (lane >> 4);
                        const float x = (ou < 0 || iu >= cols) ? 0.0f
                                                               : (float)ldexp(gate_scale(gate) * (double)wm[(gate * H + ou) * cols + iu], sw);
                        uint16_t t3[3];
                        rnn::split3_host(x, true, t3, &finite);
                        for (int term = 0; term < 3; ++term) out[((o * 3 + term) * 64 + lane) * 8 + j] = t3[term];
                    }
            scale[0] = ldexpf(1.0f, -(rnn::kStateShift + sw));
            scale[1] = ldexpf(1.0f, sw);
            return finite;
        };
        if (g == 0) {
            put_table(blob + LY.gru_ih[g], tile_ksteps(in), 3 * NB, [&](int o, int m, int ks, int qq) {
                const int gate = o / NB, ou = out_unit(H, o % NB, m), iu = tile_unit(in, ks, qq);
                return (ou < 0 || iu < 0) ? 0.0 : gate_scale(gate) * (double)w_ih[(gate * H + ou) * in + iu];
            });
        } else if (!put_split(w_ih, in, blob + LY.gru_ih[g], blob + LY.gru_scale[g] + 2)) {
            return VAD_EINVAL;
        }
        if (!put_split(w_hh, H, blob + LY.gru_hh[g], blob + LY.gru_scale[g])) return VAD_EINVAL;
        float* bt = blob + LY.gru_bias[g];
        put_bias(bt, H, [&](int u) { return rnn::kSigmoidScale * ((double)b_ih[u] + (double)b_hh[u]); });
        put_bias(bt + kBiasTile, H, [&](int u) { return rnn::kSigmoidScale * ((double)b_ih[H + u] + (double)b_hh[H + u]); });
        put_bias(bt + 2 * kBiasTile, H, [&](int u) { return rnn::kTanhScale * (double)b_ih[2 * H + u]; });
        put_bias(bt + 3 * kBiasTile, H, [&](int u) { return rnn::kTanhScale * (double)b_hh[2 * H + u]; });
    }
    for (int s = 0; s < D::ne; ++s) {                            // Linear + BatchNorm1d (+ Dropout, ReLU)
        const Affine a = affine();
        const int in = D::embed_in(s), out = D::embed_out(s), cols = s == 0 ? D::head_cols : in;
        for (int part = 0; part < (s == 0 ? D::layers : 1); ++part)
            put_table(blob + LY.emb[s] + part * LY.emb_tab[s], tile_ksteps(in), n_blocks(out), [&](int o, int m, int ks, int qq) {
                const int ou = out_unit(out, o, m), iu = tile_unit(in, ks, qq);
                return (ou < 0 || iu < 0) ? 0.0 : a.scale(ou) * (double)a.w[ou * cols + part * in + iu];
            });
        put_bias(blob + LY.emb_bias[s], out, [&](int u) { return a.shift(u); });
    }
    {
        const float *w = t[idx], *b = t[idx + 1];                // the last Linear: (2, 18)
        idx += 2;
        const int in = D::embed_in(D::ne);
        put_table(blob + LY.emb[D::ne], tile_ksteps(in), 1, [&](int, int m, int ks, int qq) {
            const int ou = out_unit(2, 0, m), iu = tile_unit(in, ks, qq);
            return (ou < 0 || iu < 0) ? 0.0 : (double)w[ou * in + iu];
        });
        put_bias(blob + LY.emb_bias[D::ne], 2, [&](int u) { return (double)b[u]; });
    }
    return idx == D::n_tensors ? VAD_OK : VAD_EINVAL;
}

template <bool LARGE>
int launch(const float* x, long n_windows, WindowMap wm, const float* trunk_in, const float* blob, float* p, uint8_t* dec,
           int mode, float threshold, StageOut so, hipStream_t st) {
    constexpr NnLayout LY = nn_layout<LARGE>();
    int dev = 0, cus = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long per_block = (long)kWaves * kWaveWindows;
    long blocks = (n_windows + per_block - 1) / per_block;
    if (blocks > cus) blocks = cus;                              // one block per CU (LDS); its waves stride over the tiles
    const bool stages = so.trunk || so.seq || so.hn || so.logits;
    if (!trunk_in && !wm.rows && wm.stride == kF && !stages) {   // stride-1 windows of a packed feature matrix
        return launch_stream(LARGE, x, n_windows, blob, p, dec, mode, threshold, (int)blocks, st);
    }
    hipLaunchKernelGGL(nicknet_kernel<LARGE>, dim3((unsigned)blocks), dim3(kWaves * 64), (size_t)LY.lds * sizeof(float), st, x,
                       n_windows, wm, trunk_in, blob, p, dec, mode, threshold, so);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

int forward(const float* x_dev, int64_t n_windows, WindowMap wm, const float* trunk_in, const float* blob_dev, int large,
            float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold, StageOut so, void* stream) {
    VAD_REQUIRE((x_dev || trunk_in) && blob_dev, "null pointer argument");
    VAD_REQUIRE(n_windows >= 0, "n_windows must be >= 0");
    VAD_REQUIRE(wm.stride >= 0 && (wm.stride % 4) == 0, "window_stride must be a non-negative multiple of 4 floats");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0, "x_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(blob_dev) & 15) == 0, "blob_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(p_dev) & 7) == 0 && (reinterpret_cast<uintptr_t>(so.logits) & 7) == 0,
                "p_dev and logits_dev must be 8-byte aligned");
    VAD_REQUIRE(decision_mode == VAD_DECIDE_ARGMAX || decision_mode == VAD_DECIDE_THRESHOLD, "unknown decision_mode %d",
                decision_mode);
    if (n_windows == 0) return VAD_OK;
    int rc = ensure_nicknet_attrs();
    if (rc != VAD_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return large ? launch<true>(x_dev, n_windows, wm, trunk_in, blob_dev, p_dev, decision_dev, decision_mode, threshold, so, st)
                 : launch<false>(x_dev, n_windows, wm, trunk_in, blob_dev, p_dev, decision_dev, decision_mode, threshold, so, st);
}

}  // namespace nn

// The kernels use more than 64 KB of dynamic LDS: opt in once per device.
int ensure_nicknet_attrs() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    VAD_HIP_CHECK(hipGetDevice(&dev));
    VAD_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return VAD_OK;
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nn::nicknet_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, nn::nn_layout<true>().lds * (int)sizeof(float)));
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nn::nicknet_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, nn::nn_layout<false>().lds * (int)sizeof(float)));
    const int rc = nn::stream_attrs();
    if (rc != VAD_OK) return rc;
    done[dev] = true;
    return VAD_OK;
}

}  // namespace vad

extern "C" size_t vad_nicknet_blob_floats(int large) {
    return (size_t)(large ? vad::nn::nn_layout<true>().total : vad::nn::nn_layout<false>().total);
}

extern "C" int vad_nicknet_tensor_count(int large) { return large ? vad::nn::Dims<true>::n_tensors : vad::nn::Dims<false>::n_tensors; }

extern "C" int vad_pack_nicknet_weights(int large, const float* const* tensors_host, int n_tensors, float* blob_host,
                                        size_t blob_floats) {
    using namespace vad;
    VAD_REQUIRE(tensors_host && blob_host, "null pointer argument");
    VAD_REQUIRE(n_tensors == vad_nicknet_tensor_count(large), "expected %d tensors, got %d", vad_nicknet_tensor_count(large),
                n_tensors);
    for (int i = 0; i < n_tensors; ++i) VAD_REQUIRE(tensors_host[i], "tensor %d is null", i);
    VAD_REQUIRE(blob_floats >= vad_nicknet_blob_floats(large), "blob too small: need %zu floats",
                vad_nicknet_blob_floats(large));
    return large ? nn::pack<true>(tensors_host, blob_host) : nn::pack<false>(tensors_host, blob_host);
}

extern "C" int vad_nicknet_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride, const float* blob_dev,
                                       int large, float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                                       void* stream) {
    using namespace vad;
    VAD_REQUIRE(x_dev && p_dev, "null pointer argument");
    return nn::forward(x_dev, n_windows, WindowMap{(long)window_stride, nullptr, 0}, nullptr, blob_dev, large, p_dev,
                       decision_dev, decision_mode, threshold, nn::StageOut{nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int vad_nicknet_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                            int64_t n_windows, const float* blob_dev, int large, float* p_dev,
                                            uint8_t* decision_dev, int decision_mode, float threshold, void* stream) {
    using namespace vad;
    VAD_REQUIRE(feat_dev && p_dev, "null pointer argument");
    WindowMap wm;
    const int rc = window_map_for_rows(n_rows, window_rows_dev, n_windows, &wm);
    if (rc != VAD_OK) return rc;
    return nn::forward(feat_dev, n_windows, wm, nullptr, blob_dev, large, p_dev, decision_dev, decision_mode, threshold,
                       nn::StageOut{nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int vad_nicknet_stages_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                                      const float* trunk_in_dev, const float* blob_dev, int large, float* trunk_out_dev,
                                      float* seq_out_dev, float* hn_out_dev, float* logits_out_dev, float* p_dev,
                                      void* stream) {
    using namespace vad;
    VAD_REQUIRE((x_dev != nullptr) != (trunk_in_dev != nullptr), "exactly one of x_dev and trunk_in_dev must be given");
    VAD_REQUIRE(!(trunk_in_dev && trunk_out_dev), "trunk_out_dev needs x_dev");
    return nn::forward(x_dev, n_windows, WindowMap{(long)window_stride, nullptr, 0}, trunk_in_dev, blob_dev, large, p_dev,
                       nullptr, VAD_DECIDE_ARGMAX, 0.0f, nn::StageOut{trunk_out_dev, seq_out_dev, hn_out_dev, logits_out_dev},
                       stream);
}

extern "C" int vad_nicknet_embed_f32(const float* v_dev, int64_t n_rows, const float* blob_dev, int large,
                                     float* logits_dev, float* p_dev, void* stream) {
    using namespace vad;
    VAD_REQUIRE(v_dev && blob_dev && (logits_dev || p_dev), "null pointer argument");
    VAD_REQUIRE(n_rows >= 0, "n_rows must be >= 0");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(blob_dev) & 15) == 0, "blob_dev must be 16-byte aligned");
    VAD_REQUIRE((reinterpret_cast<uintptr_t>(logits_dev) & 7) == 0 && (reinterpret_cast<uintptr_t>(p_dev) & 7) == 0,
                "logits_dev and p_dev must be 8-byte aligned");
    if (n_rows == 0) return VAD_OK;
    const long blocks = (n_rows + 63) / 64;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL, "too many rows for one launch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (large)
        hipLaunchKernelGGL(nn::nicknet_embed_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, v_dev, (long)n_rows, blob_dev,
                           logits_dev, p_dev);
    else
        hipLaunchKernelGGL(nn::nicknet_embed_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, v_dev, (long)n_rows, blob_dev,
                           logits_dev, p_dev);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}
