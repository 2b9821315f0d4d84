// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment: with a = 1 in one lane and b = 1 in
// one lane the product is non-zero only when the two lanes carry the same k, and then the one non-zero
// accumulator element tells which (row, column) the pair of lanes addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int* out) {       // out[la][lb] = lane * 4 + reg of the non-zero element, or -1
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            f64x4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.0) out[la * 64 + lb] = lane * 4 + r;
        }
}

int main() {
    int* d;
    static int h[64 * 64];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0xFF, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // a-lane la pairs with b-lanes of the same k: print for la = 0..63 the first matching lb and the hit
    for (int la = 0; la < 64; la += 1) {
        printf("a-lane %2d:", la);
        int shown = 0;
        for (int lb = 0; lb < 64 && shown < 3; ++lb)
            if (h[la * 64 + lb] >= 0) { printf("  b-lane %2d -> lane %2d reg %d", lb, h[la * 64 + lb] / 4, h[la * 64 + lb] % 4); ++shown; }
        printf("\n");
    }
    return 0;
}
