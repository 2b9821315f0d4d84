/*
 * vad_amd.h - C ABI of libvad_amd.so, the MI355X (gfx950) implementation of the
 * voice-activity-detection inference hot path:
 *
 *     int16 audio frames -> 12 MFCC + 12 delta per 30 ms frame
 *                        -> 30-frame (900 ms) windows
 *                        -> GRU / LSTM / DenseNet forward -> 2-class posterior
 *                        -> thresholded decision
 *
 * Conventions
 *   - Every pointer named *_dev is a DEVICE pointer owned by the caller; every
 *     pointer named *_host is a host pointer.  Nothing is allocated or freed
 *     on behalf of the caller except the per-device constant tables created by
 *     vad_init().
 *   - Kernels are enqueued on `stream` (a hipStream_t passed as void*, NULL =
 *     the default stream) and the call returns without synchronising.
 *   - Return value: 0 on success, a negative VAD_E* code otherwise; no C++
 *     exception crosses the boundary.  vad_last_error() returns a
 *     thread-local, human-readable message for the last failure.
 *   - The reference implementation is pure Python; each entry point cites the
 *     reference function whose arithmetic it replaces (paths relative to the
 *     reference repository root).
 */
#ifndef VAD_AMD_H
#define VAD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAD_OK          0
#define VAD_EINVAL     -1   /* bad argument (null pointer, misalignment, size out of range) */
#define VAD_EHIP       -2   /* a HIP runtime call failed; see vad_last_error() */
#define VAD_ENOTINIT   -3   /* vad_init() has not succeeded on the current device */

/* Constants of the path (core/common.py:17-23, core/process_data.py:48-56). */
#define VAD_SAMPLE_RATE     16000
#define VAD_FRAME_SIZE      480    /* samples per 30 ms frame */
#define VAD_ALIGN_FRAMES    3      /* MFCC_WINDOW_FRAME_SIZE - 1 */
#define VAD_NUM_CEPS        12     /* c1..c12 (c0 dropped, core/process_data.py:276) */
#define VAD_FEATURES        24     /* 12 MFCC || 12 delta */
#define VAD_FRAMES          30     /* frames per model window */

/* Decision rule written next to the posterior. */
#define VAD_DECIDE_ARGMAX     0    /* torch.max(out,1)[1]          core/train.py:605-607 */
#define VAD_DECIDE_THRESHOLD  1    /* 1 if out[:,1] >= t else 0     core/train.py:486-487 */

/* Version / liveness probe; never touches the GPU. */
int vad_abi_version(void);
const char* vad_last_error(void);

/*
 * Build the constant tables (FFT twiddles, mel segment map, DCT/lifter matrix)
 * for the CURRENT HIP device.  Idempotent; the compute entry points call it
 * lazily, but a caller that captures HIP graphs must call it beforehand.
 */
int vad_init(void);

/*
 * MFCC + delta front end.  Replaces, for a batch of slices,
 *     python_speech_features.mfcc(frames_aligned, 16000, winstep=.03, winlen=.12, nfft=2048)[:, 1:]
 *     python_speech_features.delta(mfcc, 2)
 * as called by process_test_frames (core/process_data.py:257-279) and by the
 * MFCC tail of add_noise (core/process_data.py:299-309).
 *
 * frames_dev            int16 samples.  Slice s occupies
 *                       frames_dev[s*slice_stride .. s*slice_stride + (n_s+3)*480):
 *                       3 alignment rows followed by its n_s data rows
 *                       (the concatenation at core/process_data.py:270).
 *                       slice_stride = frames_per_slice*480 describes one
 *                       contiguous stream cut into slices (each slice's
 *                       alignment rows are the previous slice's last rows,
 *                       core/process_data.py:228-229); (frames_per_slice+3)*480
 *                       describes independently stored slices.
 * n_slices              number of slices S >= 1.
 * frames_per_slice      n_s for all slices but the last.
 * last_slice_frames     n_s of the last slice, 1..frames_per_slice.
 * mfcc_dev / mfcc_ld    output row g = s*frames_per_slice + j gets its 12
 *                       coefficients at mfcc_dev[g*mfcc_ld .. +12).
 * delta_dev / delta_ld  same for the deltas.  Passing mfcc_ld = delta_ld = 24
 *                       and delta_dev = mfcc_dev + 12 produces the packed
 *                       (F,24) feature matrix the models consume
 *                       (core/generator.py:145-148).
 *
 * Per-slice semantics are kept: the first alignment row of a slice is not
 * pre-emphasised and deltas replicate the slice's own edge rows.
 * frames_dev must be 4-byte aligned and slice_stride even.
 */
int vad_mfcc_delta_f32(const int16_t* frames_dev, int64_t n_slices, int64_t frames_per_slice,
                       int64_t last_slice_frames, int64_t slice_stride,
                       float* mfcc_dev, int64_t mfcc_ld, float* delta_dev, int64_t delta_ld,
                       void* stream);

/*
 * The same front end over ONE contiguous stream cut into slices of different lengths: the loop of
 * process_training_data (core/process_data.py:123-158), which calls add_noise once per randomly sized
 * slice (33..166 frames, :88, :99) with the previous 3 frames of the same output stream as alignment rows
 * (:145), becomes one launch per noise level.
 *
 * frames_dev         int16, (n_frames + 3) * 480 samples: 3 alignment rows (zeros for the start of a
 *                    recording, :143), then the n_frames data rows; slice s covers rows
 *                    slice_starts[s] .. slice_starts[s+1] - 1 and its alignment rows are the 3 rows before.
 * slice_starts_dev   n_slices + 1 ascending int64 on the device, [0] = 0, [n_slices] = n_frames;
 *                    empty slices are allowed.  The table is device memory and is NOT validated: entries are
 *                    clamped into 0 .. n_frames (and a slice to 2^30 frames), so a table that is not ascending or
 *                    does not end at n_frames leaves output rows unwritten or cut differently, but cannot make
 *                    the kernel touch memory outside the buffers described here.
 * max_slice_frames   an upper bound of the slice lengths (sizes the launch: one wave per slice up to 256
 *                    frames).  A longer slice is still computed in full, only less evenly spread.
 * Outputs and per-slice semantics as vad_mfcc_delta_f32: output row g = frame g of the stream.
 */
int vad_mfcc_delta_slices_f32(const int16_t* frames_dev, const int64_t* slice_starts_dev, int64_t n_slices,
                              int64_t n_frames, int64_t max_slice_frames,
                              float* mfcc_dev, int64_t mfcc_ld, float* delta_dev, int64_t delta_ld, void* stream);

/*
 * Weight packing (host side, no GPU work).  State-dict tensors of the
 * reference module -> one float blob in the layout the kernels read
 * (MFMA A-operand order, BatchNorm folded).  The caller copies the blob to
 * the device.
 *
 * Net (core/models.py:9-81): rnn.weight_ih_l0 (G*H,24), rnn.weight_hh_l0
 * (G*H,H), rnn.bias_ih_l0, rnn.bias_hh_l0 (G*H) with G = 3 (GRU: r,z,n) or
 * 4 (LSTM: i,f,g,o); lin1 (26, 30*H) + (26); lin2 (2,26) + (2); lin (2, 30*H)
 * + (2).  `large` selects lin1/lin2, otherwise lin; the unused head may be
 * NULL.  hidden = 30 is the reference architecture; 1..32 is accepted for the
 * LSTM and 1..96 for the GRU (the enlarged-GRU configuration uses 90).
 */
size_t vad_net_blob_floats(int lstm, int large, int hidden);
int vad_pack_net_weights(int lstm, int large, int hidden,
                         const float* w_ih_host, const float* w_hh_host,
                         const float* b_ih_host, const float* b_hh_host,
                         const float* lin1_w_host, const float* lin1_b_host,
                         const float* lin2_w_host, const float* lin2_b_host,
                         const float* lin_w_host, const float* lin_b_host,
                         float* blob_host, size_t blob_floats);

/*
 * DenseNet (core/models.py:383-461).  `tensors_host` lists, in state_dict()
 * order and skipping every num_batches_tracked entry, the float tensors of
 * the module: for each Conv1d+BatchNorm1d pair conv.weight, bn.weight,
 * bn.bias, bn.running_mean, bn.running_var; finally out.weight.
 * That is 5*(1 + L + 1 + L + 1) + 1 tensors with L = 6 (small) or 8 (large).
 * BatchNorm uses eps = 1e-5 and is folded into the convolutions.
 */
size_t vad_densenet_blob_floats(int large);
int vad_densenet_tensor_count(int large);
int vad_pack_densenet_weights(int large, const float* const* tensors_host, int n_tensors,
                              float* blob_host, size_t blob_floats);

/*
 * Model forward over windows.  Replaces Net.forward (core/models.py:39-81)
 * and DenseNet.forward (core/models.py:449-461) in eval mode, plus the
 * decision step of netvad / far.apply_threshold.
 *
 * x_dev           float features.  Window w reads its 30 rows of 24 floats at
 *                 x_dev[w*window_stride + t*24 .. +24), t = 0..29.
 *                 window_stride = 720 is a dense (B,30,24) batch as passed to
 *                 forward(); window_stride = 24 is the stride-1 sliding window
 *                 over a packed (F,24) feature matrix (core/train.py:566-570),
 *                 in which case n_windows = F - 29; 24*k gives stride k.
 *                 Must be 16-byte aligned, window_stride a multiple of 4.
 * blob_dev        device copy of the packed weights.
 * p_dev           (n_windows, 2) softmax posteriors; column 1 = speech.
 * decision_dev    optional (may be NULL) uint8 per window.
 * decision_mode   VAD_DECIDE_ARGMAX or VAD_DECIDE_THRESHOLD (uses `threshold`).
 */
int vad_gru_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                        const float* blob_dev, int hidden, int large,
                        float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                        void* stream);
int vad_lstm_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                         const float* blob_dev, int hidden, int large,
                         float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                         void* stream);
int vad_densenet_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                             const float* blob_dev, int large,
                             float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                             void* stream);

/*
 * vad_gru_forward_f32 with a workspace, for GRUs with more than 32 hidden units (the enlarged ~100k-parameter
 * configuration, hidden = 90; no class in the reference) over stride-1 windows (window_stride == 24).  The head
 * of every step's accumulation chain, bias + W_ih x_row, depends on the feature row only; with a workspace
 * it is computed once per row instead of once per window and step (the recurrent weights of such a model fill
 * the LDS, so unlike the 30-unit models' kernel it cannot keep the projected rows on chip).  In the
 * VAD_MATRIX_F32 mode the results are bit-identical to vad_gru_forward_f32; in the bf16 matrix modes (below) the
 * workspace path forms the recurrent and head products on the bf16 matrix pipe like the 30-unit models' stride-1
 * kernels.  Other models and strides ignore the workspace.
 *   workspace_dev    caller-owned device memory, 16-byte aligned; any size: the windows are processed in chunks
 *                    of as many as the workspace has rows for (288 floats per row for hidden 65..96, 192 for
 *                    33..64; at least 30 rows, else the workspace is not used).  Contents are scratch.
 * vad_gru_workspace_bytes: the size that takes n_windows windows in one chunk (0 where no workspace is used).
 */
size_t vad_gru_workspace_bytes(int hidden, int64_t n_windows);
int vad_gru_forward_ws_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                           const float* blob_dev, int hidden, int large,
                           float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/*
 * How the stride-1 (window_stride == 24) GRU / LSTM kernels with hidden <= 32, and the wider GRUs' stride-1
 * workspace path (vad_gru_forward_ws_f32), form the float32 products of the recurrence and of the 26-row head.
 * Process-wide default (vad_set_matrix_mode) or per call (vad_*_forward_mode_f32); no counterpart in the reference.
 *   VAD_MATRIX_BF16X9  every float32 operand is split exactly into three bfloat16 (x = hi + mid + lo)
 *                      and all nine partial products, each exact in float32, are accumulated in float32 on the
 *                      bf16 matrix pipe (v_mfma_f32_16x16x32_bf16): no operand or product is rounded, only the
 *                      accumulation is, as in a float32 FMA chain but in a different order and with the matrix
 *                      pipe's alignment: the instruction adds its K = 32 in four chained groups of 8; in each the
 *                      exact products are aligned to the largest one and kept to 25 bits before the sum is added
 *                      to C and rounded to nearest (measured by tools/probes/mfma_bf16_accum_k32.hip), which
 *                      bounds a K = 32 dot product's error
 *                      by about 68 * 2^-24 * the largest magnitude in the chain (float32 FMA chain: 30 * 2^-24).
 *   VAD_MATRIX_BF16X8  as BF16X9 without the lo * lo partial product: the operands are split by
 *                      truncation, so |lo| < 2^-15 |x| and the dropped term is below 2^-30 of the product it
 *                      belongs to, 1/64 of the rounding unit of a single float32 operation; it adds
 *                      2^-30 * sum |a b| (under 2^-6 * 2^-24 of the chain's magnitude per term) to the bound above.
 *   VAD_MATRIX_F32     v_mfma_f32_32x32x2_f32: a float32 FMA chain in k order, bit-identical to what every other
 *                      path of this library computes (dense batches, other strides, sampled windows).
 *   VAD_MATRIX_BF16X6  as BF16X9 without the three smallest partial products (<= 2^-23 of a product).
 *   VAD_MATRIX_F16X6   (default) the same construction on float16 operands (v_mfma_f32_16x16x32_f16): float16 has 11 significant
 *                      bits, so x = hi + mid + lo by round-to-nearest has |mid| <= 2^-12 |x|, |lo| <= 2^-24 |x| and the six
 *                      largest partial products leave out less than 2^-35 of a product -- the reach of BF16X9 for the
 *                      instructions of BF16X6.  float16's narrow exponent is handled with powers of two, which round
 *                      nothing: the hidden state is carried as 2^14 h and the packer multiplies the weights by the
 *                      largest 2^Sw that keeps them below 65504, so that every operand of magnitude >= 0.5 after scaling
 *                      -- hidden states from 2^-15, weights from 2^-14 of the largest -- is split exactly (smaller ones are
 *                      rounded to 2^-25 of the scaled unit); the accumulators hold 2^(14+Sw) times the pre-activations and
 *                      the gate math multiplies that back out (two multiplies per GRU unit and step, four per LSTM unit).
 *   VAD_MATRIX_F16X4   hi * hi, hi * mid, mid * hi, mid * mid of the same split: products to 2^-23, four instructions.
 * Other strides, sampled windows, wider GRUs without a workspace and DenseNet always compute as VAD_MATRIX_F32.
 */
#define VAD_MATRIX_F32 0
#define VAD_MATRIX_BF16X6 1
#define VAD_MATRIX_BF16X9 2
#define VAD_MATRIX_BF16X8 3
#define VAD_MATRIX_F16X6 4
#define VAD_MATRIX_F16X4 5
#define VAD_MATRIX_DEFAULT (-1)   /* per-call argument only: use the process-wide mode */
int vad_set_matrix_mode(int mode);
int vad_get_matrix_mode(void);

/*
 * Net.forward (core/models.py:39-81) for inputs that are not (B,30,24) windows.  The reference widens a 2-D input
 * (A,24) to A sequences of one step and a 1-D input to one sequence of one step (core/models.py:44-47), runs its
 * recurrent layer over any sequence length and re-cuts ALL hidden states, in (sequence, step) order, into rows of 900
 * values (view(-1, FRAMES ** 2), core/models.py:66) before the head.  This entry point does the same for n_seq
 * sequences of seq_len steps (hidden size 30): p_dev receives n_seq * seq_len / 30 posterior rows; a product that
 * is not a multiple of 30 is VAD_EINVAL, where the reference's view() raises.  Weights are device copies of the
 * state dict in PyTorch's own layout (rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0; head:
 * lin1.weight, lin1.bias, lin2.weight, lin2.bias for large != 0, else lin.weight, lin.bias and two NULLs).
 * hseq_dev is scratch for n_seq * seq_len * 30 floats.  A plain kernel for completeness of the drop-in, not a tuned
 * one: windows go through vad_gru_forward_f32 / vad_lstm_forward_f32.
 */
int vad_net_forward_seq_f32(const float* x_dev, int64_t n_seq, int64_t seq_len,
                            const float* w_ih_dev, const float* w_hh_dev, const float* b_ih_dev, const float* b_hh_dev,
                            const float* head_w1_dev, const float* head_b1_dev,
                            const float* head_w2_dev, const float* head_b2_dev,
                            int lstm, int large, float* hseq_dev, float* p_dev, void* stream);

/*
 * vad_gru_forward_ws_f32 / vad_lstm_forward_f32 with the matrix mode as an argument of the call instead of the
 * process-wide setting, so that models (or callers, or streams) of one process can differ in it: one model
 * bit-identical to its dense-batch forward() (VAD_MATRIX_F32), another on the faster split products.
 * matrix_mode is one of the VAD_MATRIX_* values above; VAD_MATRIX_DEFAULT reads the process-wide mode
 * at the time of the call.  The GRU form takes the optional workspace of vad_gru_forward_ws_f32
 * (NULL / 0 = none).  Same Net.forward semantics (core/models.py:39-81) as the calls they extend.
 */
int vad_gru_forward_mode_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                             const float* blob_dev, int hidden, int large,
                             float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                             int matrix_mode, void* workspace_dev, size_t workspace_bytes, void* stream);
int vad_lstm_forward_mode_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                              const float* blob_dev, int hidden, int large,
                              float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                              int matrix_mode, void* stream);

/*
 * Sampled windows: window w starts at feature row window_rows_dev[w] of a packed
 * (n_rows, 24) matrix.  This is the access pattern of the reference's batch
 * provider, DataGenerator.get_batch (core/generator.py:111-165), which draws
 * class-balanced window starts at random and stacks mfcc[i:i+30] | delta[i:i+30]
 * per start (:143-150) before test_predict scores the batch (core/train.py:416-440);
 * here the batch tensor is never built.  Row indices outside 0 .. n_rows-30 are
 * clamped into that range.  Other arguments as in the *_forward_f32 calls.
 */
int vad_gru_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                             int64_t n_windows, const float* blob_dev, int hidden, int large,
                             float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold, void* stream);
int vad_lstm_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                              int64_t n_windows, const float* blob_dev, int hidden, int large,
                              float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold, void* stream);
int vad_densenet_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                  int64_t n_windows, const float* blob_dev, int large,
                                  float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold, void* stream);

/*
 * NickNet, the reference's conv + GRU hybrid (core/models.py:245-330), in its default configuration:
 * gated convolutions (GatedConv, core/models.py:180-202: tanh(BN(conv)) * sigmoid(BN(conv_gate)), kernel 3,
 * padding 1), no residual connections, uni-directional GRU and the fully connected `embed` head of BiRNN
 * (core/models.py:84-128), eval mode.  large: channels 24 -> 32 -> 28 -> 25 -> 18, GRU 18 -> 29 -> 29 (2 layers),
 * embed 58 -> 32 -> 18 -> 2; small: 24 -> 20 -> 18 -> 16, GRU 16 -> 21, embed 21 -> 18 -> 2.
 *
 * The reference's NickNet.forward raises between the GRU and `embed` (BiRNN.forward flattens all 30 GRU outputs
 * into 870 columns where `embed` takes L*H = 58 / 21, core/models.py:166-174), so the whole-model output has no
 * reference counterpart.  The conv trunk (core/models.py:308-323), the GRU (:151,162) and `embed` (:109-128)
 * do run in the reference and are what vad_nicknet_stages_f32 / vad_nicknet_embed_f32 expose for checking.
 * The head input used here is the final hidden state of every GRU layer, h_n.permute(1,0,2).reshape(B, L*H).
 * Arithmetic: float32 operands and accumulation; the convolutions, the first GRU layer's input products and the head
 * on the float32 MFMA, the GRU's other products (those of a hidden state) as six float16 partial products of exactly
 * split operands on the 16-bit MFMA (what is dropped is below 2^-35 of a product).
 *
 * `tensors_host`: the float tensors of the module in state_dict() order, skipping num_batches_tracked:
 * per conv layer conv.0.weight (cout,cin,3), conv.0.bias, conv.1.{weight,bias,running_mean,running_var} and
 * the same six of conv_gate; per GRU layer weight_ih, weight_hh, bias_ih, bias_hh; per hidden embed stage
 * Linear weight, bias and BatchNorm1d weight, bias, running_mean, running_var; the last Linear's weight, bias.
 * 70 tensors (large) or 48 (small).  BatchNorm uses eps = 1e-5 and is folded into the weights.
 */
size_t vad_nicknet_blob_floats(int large);
int vad_nicknet_tensor_count(int large);
int vad_pack_nicknet_weights(int large, const float* const* tensors_host, int n_tensors,
                             float* blob_host, size_t blob_floats);
/* Windows -> posteriors and decisions; arguments as vad_densenet_forward_f32 / vad_densenet_forward_rows_f32. */
int vad_nicknet_forward_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                            const float* blob_dev, int large,
                            float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold,
                            void* stream);
int vad_nicknet_forward_rows_f32(const float* feat_dev, int64_t n_rows, const int64_t* window_rows_dev,
                                 int64_t n_windows, const float* blob_dev, int large,
                                 float* p_dev, uint8_t* decision_dev, int decision_mode, float threshold, void* stream);
/*
 * The same forward with the outputs of its sub-modules written out in the reference's layouts (any may be NULL):
 *   trunk_out_dev   (B, C, 30)  conv1 .. conv4 of x.permute(0,2,1)                 core/models.py:308-323
 *   seq_out_dev     (B, 30, H)  outputs of the last GRU layer, net.rnn.rnn(...)[0]   core/models.py:162
 *   hn_out_dev      (L, B, H)   final hidden state of every GRU layer, net.rnn.rnn(...)[1]
 *   logits_out_dev  (B, 2)      embed output before the softmax                      core/models.py:177
 *   p_dev           (B, 2)      softmax                                              core/models.py:330
 * Exactly one of x_dev (windows, as above) and trunk_in_dev (a (B, C, 30) trunk output to start the GRU
 * from, e.g. the reference's own) is given.
 */
int vad_nicknet_stages_f32(const float* x_dev, int64_t n_windows, int64_t window_stride,
                           const float* trunk_in_dev, const float* blob_dev, int large,
                           float* trunk_out_dev, float* seq_out_dev, float* hn_out_dev, float* logits_out_dev,
                           float* p_dev, void* stream);
/* `embed` alone: rows of v_dev (n_rows, L*H) -> logits (n_rows, 2) and / or their softmax (core/models.py:109-128). */
int vad_nicknet_embed_f32(const float* v_dev, int64_t n_rows, const float* blob_dev, int large,
                          float* logits_dev, float* p_dev, void* stream);

/*
 * Noise/speech mix ahead of the feature call on the training-data path: the
 * int16 arithmetic of `noise_track.overlay(speech_track, gain_during_overlay=dB)`
 * in add_noise (core/process_data.py:283-297; pydub -> audioop.mul + audioop.add):
 *   out[i] = sat16( (apply_gain ? sat16(floor(noise[i] * gain_factor)) : noise[i]) + speech[i] ),  i < min(n_noise, n_speech)
 *   out[i] = noise[i]                                                                             for the rest of the noise
 * gain_factor = 10^(dB/20), computed by the caller in double precision as pydub
 * does; a falsy gain (None, 0) in the reference means apply_gain = 0.
 * noise_dev / speech_dev / out_dev: int16 device arrays of n_noise / n_speech /
 * n_noise samples; out_dev may alias noise_dev.  16-byte aligned pointers take the
 * wide path.
 */
int vad_overlay_i16(const int16_t* noise_dev, int64_t n_noise, const int16_t* speech_dev, int64_t n_speech,
                    int apply_gain, double gain_factor, int16_t* out_dev, void* stream);

/*
 * Threshold sweep over posteriors: the consumer side of the path.  For an
 * ascending table of thresholds t_0 < ... < t_{T-1} (T <= VAD_MAX_THRESHOLDS)
 * one pass counts, per class, the scores with score >= t_k -- the confusion
 * matrix of every threshold at once.  Replaces the per-candidate
 * `apply_threshold` (y >= t) + `confusion_matrix` of the FAR-at-fixed-FRR search
 * fix_frr (core/train.py:486-528), which rescans all scores for each t *= 1.1 step.
 * The comparison is exact: (double)score >= t_k; NaN scores are below every threshold.
 * The reference's `y >= t` compares a numpy.float32 with a Python float, which numpy >= 2
 * evaluates in float32 (t rounded to float32; numpy 1.x used float64): a caller that wants
 * that rule passes thresholds already rounded to float32, as the host side's fix_frr does.
 * The per-window decision of the forward calls (p1 >= threshold, float32) is the same rule.
 *   score_dev       float32, element i at score_dev[i * score_stride] (stride 2 reads
 *                   column 1 of an (n,2) posterior array in place)
 *   label_dev       uint8, non-zero = speech
 *   thresholds_dev  T doubles, ascending
 *   hist_dev        scratch/out, 2*(T+1) uint64: [bin][class], bin = #thresholds <= score
 *   ge_dev          out, T x 2 int64: [k][class] = scores of that class >= t_k
 *                   (class 1: true positives, class 0: false positives at threshold k)
 *   total_dev       out, 2 int64: scores per class
 */
#define VAD_MAX_THRESHOLDS 2048
int vad_threshold_sweep_f32(const float* score_dev, int64_t score_stride, const uint8_t* label_dev, int64_t n,
                            const double* thresholds_dev, int n_thresholds,
                            unsigned long long* hist_dev, int64_t* ge_dev, int64_t* total_dev, void* stream);

/*
 * The same sweep for callers that cannot allocate device memory and so have no workspace to pass
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
 * in the reference.  The (T+1) x 2 histogram lives in an array built into the library, 2 * (VAD_MAX_THRESHOLDS
 * + 1) 64-bit words = 32 784 bytes per GPU, held from the library's load until the process ends; the call
 * makes no hipMalloc / hipFree and clears the array itself.  n_thresholds > VAD_MAX_THRESHOLDS is
 * rejected with VAD_EINVAL.  All calls on one GPU share that array: calls that overlap on one GPU
 * (different streams or host threads) race; calls on one stream, or on different GPUs, do not.
 * vad_init() looks the array's device address up for the current device; a caller that captures graphs
 * calls it beforehand, as for the other entry points.  Results equal vad_threshold_sweep_f32's.
 */
int vad_threshold_sweep_noalloc_f32(const float* score_dev, int64_t score_stride, const uint8_t* label_dev, int64_t n,
                                    const double* thresholds_dev, int n_thresholds,
                                    int64_t* ge_dev, int64_t* total_dev, void* stream);

/*
 * Deterministic synthetic 16 kHz int16 audio, written on the device: sample n
 * of stream `seed` is a pure integer function of (seed, n), reproducible on
 * the host.  No counterpart in the reference (its audio comes from datasets
 * that do not ship, core/prepare_files.py); used by the benchmark and tests.
 */
int vad_synth_audio_i16(int16_t* out_dev, int64_t first_sample, int64_t n_samples,
                        uint32_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAD_AMD_H */
