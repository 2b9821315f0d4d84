// Net.forward of the reference (core/models.py:39-81) for inputs that are NOT (B,30,24) windows.
//
// The reference widens a 2-D input (A,24) to A sequences of one step and a 1-D input to one sequence of one
// step (core/models.py:44-47), runs its recurrent layer over whatever sequence length it is given and then
// re-cuts ALL hidden states, in (sequence, step) order, into rows of 900 values (`view(-1, FRAMES ** 2)`,
// core/models.py:66) before the head: a row of the output is 30 consecutive (sequence, step) hidden states,
// whichever sequences they belong to.  Nothing in the reference feeds it such shapes; this file exists so that
// forward() is a drop-in for every shape the reference's accepts.  It is a plain kernel, not a tuned one: one wave
// handles two sequences, lane = hidden unit, weights in their PyTorch layout (no packed blob).

#include <math.h>

#include "vad_internal.h"

namespace vad {
namespace {

constexpr int kH = VAD_FRAMES;            // hidden units of the reference's recurrent layer (core/models.py:19-25)
constexpr int kF = VAD_FEATURES;
constexpr int kRow = VAD_FRAMES * VAD_FRAMES;      // 900 values per head input row

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + expf(-a)); }

// hseq[(s * T + t) * 30 + u] = hidden state of sequence s after step t
template <bool LSTM>
__global__ __launch_bounds__(64) void seq_rnn_kernel(const float* __restrict__ x, long n_seq, long seq_len,
                                                     const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                     const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                     float* __restrict__ hseq) {
    constexpr int G = LSTM ? 4 : 3;
    __shared__ float s_wih[G * kH * kF];
    __shared__ float s_whh[G * kH * kH];
    __shared__ float s_x[2][kF];
    __shared__ float s_h[2][32];
    for (int i = threadIdx.x; i < G * kH * kF; i += 64) s_wih[i] = w_ih[i];
    for (int i = threadIdx.x; i < G * kH * kH; i += 64) s_whh[i] = w_hh[i];
    const int half = threadIdx.x >> 5, u = threadIdx.x & 31;
    const long s = (long)blockIdx.x * 2 + half;
    const bool live = s < n_seq && u < kH;
    float bi[G], bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        bi[g] = live ? b_ih[g * kH + u] : 0.0f;
        bh[g] = live ? b_hh[g * kH + u] : 0.0f;
    }
    float h = 0.0f, c = 0.0f;
    s_h[half][u] = 0.0f;
    __syncthreads();
    for (long t = 0; t < seq_len; ++t) {
        if (s < n_seq && u < kF) s_x[half][u] = x[(s * seq_len + t) * kF + u];
        __syncthreads();
        float gi[G], gh[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            gi[g] = bi[g];
            gh[g] = bh[g];
        }
        if (live) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float* wi = s_wih + (g * kH + u) * kF;
                for (int k = 0; k < kF; ++k) gi[g] = fmaf(wi[k], s_x[half][k], gi[g]);
                const float* wh = s_whh + (g * kH + u) * kH;
                for (int j = 0; j < kH; ++j) gh[g] = fmaf(wh[j], s_h[half][j], gh[g]);
            }
            if (LSTM) {                                    // gate order i, f, g, o (torch.nn.LSTM)
                const float ig = sigmoidf_(gi[0] + gh[0]), fg = sigmoidf_(gi[1] + gh[1]);
                const float gg = tanhf(gi[2] + gh[2]), og = sigmoidf_(gi[G - 1] + gh[G - 1]);
                c = fg * c + ig * gg;
                h = og * tanhf(c);
            } else {                                       // gate order r, z, n (torch.nn.GRU)
                const float r = sigmoidf_(gi[0] + gh[0]), z = sigmoidf_(gi[1] + gh[1]);
                const float n = tanhf(gi[2] + r * gh[2]);
                h = (1.0f - z) * n + z * h;
            }
            hseq[(s * seq_len + t) * kH + u] = h;
        }
        __syncthreads();                                   // every lane has read the old state
        s_h[half][u] = h;
    }
}

// One block per output row: lin1 + ReLU + lin2 (large) or lin (small), softmax (core/models.py:74-81).
template <bool LARGE>
__global__ __launch_bounds__(64) void seq_head_kernel(const float* __restrict__ hseq, long n_rows, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ p) {
    __shared__ float s_v[kRow];
    __shared__ float s_a[32];
    const long r = blockIdx.x;
    for (int i = threadIdx.x; i < kRow; i += 64) s_v[i] = hseq[r * kRow + i];
    __syncthreads();
    const int j = threadIdx.x;
    constexpr int kOut1 = LARGE ? 26 : 2;
    if (j < kOut1) {
        float acc = b1[j];
        const float* w = w1 + (long)j * kRow;
        for (int i = 0; i < kRow; ++i) acc = fmaf(w[i], s_v[i], acc);
        s_a[j] = LARGE ? fmaxf(acc, 0.0f) : acc;
    }
    __syncthreads();
    if (j == 0) {
        float l0, l1;
        if (LARGE) {
            l0 = b2[0];
            l1 = b2[1];
            for (int k = 0; k < 26; ++k) {
                l0 = fmaf(w2[k], s_a[k], l0);
                l1 = fmaf(w2[26 + k], s_a[k], l1);
            }
        } else {
            l0 = s_a[0];
            l1 = s_a[1];
        }
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        p[2 * r] = e0 / (e0 + e1);
        p[2 * r + 1] = e1 / (e0 + e1);
    }
}

}  // namespace
}  // namespace vad

extern "C" int vad_net_forward_seq_f32(const float* x_dev, int64_t n_seq, int64_t seq_len, const float* w_ih_dev,
                                       const float* w_hh_dev, const float* b_ih_dev, const float* b_hh_dev,
                                       const float* head_w1_dev, const float* head_b1_dev, const float* head_w2_dev,
                                       const float* head_b2_dev, int lstm, int large, float* hseq_dev, float* p_dev,
                                       void* stream) {
    using namespace vad;
    VAD_REQUIRE(n_seq >= 0 && seq_len >= 0, "negative sequence count or length");
    VAD_REQUIRE((n_seq * seq_len) % VAD_FRAMES == 0,
                "%lld hidden states cannot be cut into rows of %d (the reference's view(-1, %d) fails the same way)",
                (long long)(n_seq * seq_len), VAD_FRAMES, VAD_FRAMES * VAD_FRAMES);
    if (n_seq == 0 || seq_len == 0) return VAD_OK;
    VAD_REQUIRE(x_dev && w_ih_dev && w_hh_dev && b_ih_dev && b_hh_dev && head_w1_dev && head_b1_dev && hseq_dev && p_dev,
                "null pointer");
    VAD_REQUIRE(!large || (head_w2_dev && head_b2_dev), "the large head needs lin2's weight and bias");
    const long blocks = (n_seq + 1) / 2, rows = n_seq * seq_len / VAD_FRAMES;
    VAD_REQUIRE(blocks <= 0x7FFFFFFFL && rows <= 0x7FFFFFFFL, "too many sequences for one launch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (lstm)
        hipLaunchKernelGGL((seq_rnn_kernel<true>), dim3((unsigned)blocks), dim3(64), 0, st, x_dev, n_seq, seq_len, w_ih_dev,
                           w_hh_dev, b_ih_dev, b_hh_dev, hseq_dev);
    else
        hipLaunchKernelGGL((seq_rnn_kernel<false>), dim3((unsigned)blocks), dim3(64), 0, st, x_dev, n_seq, seq_len, w_ih_dev,
                           w_hh_dev, b_ih_dev, b_hh_dev, hseq_dev);
    VAD_HIP_CHECK(hipGetLastError());
    if (large)
        hipLaunchKernelGGL((seq_head_kernel<true>), dim3((unsigned)rows), dim3(64), 0, st, hseq_dev, rows, head_w1_dev,
                           head_b1_dev, head_w2_dev, head_b2_dev, p_dev);
    else
        hipLaunchKernelGGL((seq_head_kernel<false>), dim3((unsigned)rows), dim3(64), 0, st, hseq_dev, rows, head_w1_dev,
                           head_b1_dev, head_w2_dev, head_b2_dev, p_dev);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}
