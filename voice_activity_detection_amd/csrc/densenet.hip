// DenseNet window classifier (core/models.py:383-461) -- placeholder until the kernel lands.
#include "vad_internal.h"

extern "C" size_t vad_densenet_blob_floats(int large) { (void)large; return 0; }
extern "C" int vad_densenet_tensor_count(int large) { return 5 * (3 + 2 * (large ? 8 : 6)) + 1; }
extern "C" int vad_pack_densenet_weights(int, const float* const*, int, float*, size_t) {
    vad::set_error("DenseNet kernel not built yet");
    return VAD_EINVAL;
}
extern "C" int vad_densenet_forward_f32(const float*, int64_t, int64_t, const float*, int, float*, uint8_t*, int, float,
                                        void*) {
    vad::set_error("DenseNet kernel not built yet");
    return VAD_EINVAL;
}
