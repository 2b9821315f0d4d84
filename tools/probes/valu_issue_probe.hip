// How many cycles does a SIMD spend per wave64 v_fma_f32 / v_add_f32 / v_fmac_f32 when 1, 2, 3 or 4 waves share it?
// One block per CU (100 KB of dynamic LDS keeps a second block out), block = 256 / 512 / 1024 threads.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MFMA>
__global__ void probe(float* out, int iters, float a, float b) {
    extern __shared__ float lds[];
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // inline asm keeps the compiler from pairing these into v_pk_fma_f32
            if (MFMA == 0)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (MFMA == 1)                      // VOP2 add
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (MFMA == 3)                      // SDWA conversion of a 16-bit half
            asm volatile("v_cvt_f32_i32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_i32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_i32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                         "v_cvt_f32_i32_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (MFMA == 4)                      // plain conversion
            asm volatile("v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %1, %8\n v_cvt_f32_i32 %2, %8\n v_cvt_f32_i32 %3, %8\n"
                         "v_cvt_f32_i32 %4, %8\n v_cvt_f32_i32 %5, %8\n v_cvt_f32_i32 %6, %8\n v_cvt_f32_i32 %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (MFMA == 5)                      // transcendental
            asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n"
                         "v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (MFMA == 6)                      // VOP3 select
            asm volatile("v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n"
                         "v_cndmask_b32_e64 %4, %8, %9, vcc\n v_cndmask_b32_e64 %5, %8, %9, vcc\n v_cndmask_b32_e64 %6, %8, %9, vcc\n v_cndmask_b32_e64 %7, %8, %9, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            else                                     // VOP2 fmac
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
    }
    if (iters < 0) lds[threadIdx.x] = x0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;                       // 64 FMAs per iteration per wave
    for (int op = 0; op < 7; ++op)
    for (int threads : {256, 512, 768, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            if (op == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double inst_per_simd = (double)iters * 64 * (threads / 256);
            printf("%s waves/SIMD %d  %.3f ms  %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", op == 0 ? "v_fma_f32 (VOP3)" : op == 1 ? "v_add_f32 (VOP2)" : op == 2 ? "v_fmac_f32 (VOP2)" : op == 3 ? "v_cvt_f32_i32_sdwa" : op == 4 ? "v_cvt_f32_i32" : op == 5 ? "v_exp_f32" : "v_cndmask_b32_e64", threads / 256, ms,
                   ms * 1e6 / inst_per_simd);
        }
    }
    return 0;
}
