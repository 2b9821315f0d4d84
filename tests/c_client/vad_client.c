/* A plain-C client of libvad_amd.so: no Python, no torch.  Generates a synthetic stream on the device,
 * runs the front end and the GRU over every stride-1 window, copies the posteriors back and prints a
 * summary that tests/test_gpu_c_client.py compares with the Python host side's results for the same
 * weights (read from a raw float32 blob file written by the test).
 *
 *   vad_client <blob.f32> <n_frames> <seed>
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include <hip/hip_runtime_api.h>

#include "vad_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_VAD(x) do { int r_ = (x); if (r_ != VAD_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, vad_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s blob.f32 n_frames seed\n", argv[0]); return 1; }
    const int64_t n_frames = atoll(argv[2]);
    const uint32_t seed = (uint32_t)strtoul(argv[3], NULL, 10);
    if (n_frames < VAD_FRAMES) { fprintf(stderr, "need at least %d frames\n", VAD_FRAMES); return 1; }

    const size_t blob_floats = vad_net_blob_floats(0, 1, 30);
    float* blob_host = (float*)malloc(blob_floats * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(blob_host, sizeof(float), blob_floats, f) != blob_floats) { fprintf(stderr, "cannot read %zu floats from %s\n", blob_floats, argv[1]); return 1; }
    fclose(f);

    CHECK_VAD(vad_init());
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    const int64_t n_windows = n_frames - VAD_FRAMES + 1;
    const int64_t n_samples = (n_frames + VAD_ALIGN_FRAMES) * VAD_FRAME_SIZE;
    int16_t* audio;
    float *feat, *blob, *post;
    uint8_t* dec;
    CHECK_HIP(hipMalloc((void**)&audio, n_samples * sizeof(int16_t)));
    CHECK_HIP(hipMalloc((void**)&feat, n_frames * VAD_FEATURES * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&blob, blob_floats * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&post, n_windows * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dec, n_windows));
    CHECK_HIP(hipMemcpyAsync(blob, blob_host, blob_floats * sizeof(float), hipMemcpyHostToDevice, stream));

    /* 3 zero alignment rows, then the stream (as the first slice of a recording has) */
    CHECK_HIP(hipMemsetAsync(audio, 0, VAD_ALIGN_FRAMES * VAD_FRAME_SIZE * sizeof(int16_t), stream));
    CHECK_VAD(vad_synth_audio_i16(audio + VAD_ALIGN_FRAMES * VAD_FRAME_SIZE, 0, n_frames * VAD_FRAME_SIZE, seed, stream));

    const int64_t slice = 166;
    const int64_t n_slices = (n_frames + slice - 1) / slice;
    CHECK_VAD(vad_mfcc_delta_f32(audio, n_slices, slice, n_frames - (n_slices - 1) * slice, slice * VAD_FRAME_SIZE,
                                 feat, VAD_FEATURES, feat + VAD_NUM_CEPS, VAD_FEATURES, stream));
    CHECK_VAD(vad_gru_forward_f32(feat, n_windows, VAD_FEATURES, blob, 30, 1, post, dec, VAD_DECIDE_ARGMAX, 0.0f, stream));

    float* post_host = (float*)malloc(n_windows * 2 * sizeof(float));
    uint8_t* dec_host = (uint8_t*)malloc(n_windows);
    CHECK_HIP(hipMemcpyAsync(post_host, post, n_windows * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(dec_host, dec, n_windows, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    double sum = 0.0;
    long speech = 0;
    for (int64_t w = 0; w < n_windows; ++w) { sum += post_host[2 * w + 1]; speech += dec_host[w]; }
    /* bit patterns of a few posteriors, for an exact comparison */
    printf("windows %lld sum_p1 %.17g speech %ld", (long long)n_windows, sum, speech);
    for (int64_t w = 0; w < n_windows; w += (n_windows - 1) / 4) {
        union { float f; uint32_t u; } v;
        v.f = post_host[2 * w + 1];
        printf(" %08x", v.u);
    }
    printf("\n");

    /* the error channel answers without touching the device */
    if (vad_gru_forward_f32(NULL, 1, 720, blob, 30, 1, post, NULL, 0, 0.0f, stream) != VAD_EINVAL) return 4;
    return 0;
}
