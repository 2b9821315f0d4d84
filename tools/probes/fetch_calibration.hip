// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses:
// kernels that move a KNOWN number of bytes (1 GiB, far beyond the 256 MiB Infinity Cache) with 4-byte and
// 16-byte accesses per lane.  Run once per counter:
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calibration
//     rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -- ./fetch_calibration
// and compare the counter (KiB) with 1 048 576 KiB per kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_b32_kernel(const int* __restrict__ src, long n, int* __restrict__ sink) {
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read_b128_kernel(const int4* __restrict__ src, long n, int* __restrict__ sink) {
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write_b32_kernel(int* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (int)i;
}

__global__ __launch_bounds__(256) void write_b128_kernel(int4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = make_int4((int)i, 1, 2, 3);
}

// 48 of every 64 lanes store one float of a 12-float row at a 96-byte pitch: the front end's output pattern
__global__ __launch_bounds__(256) void write_rows12_kernel(float* __restrict__ dst, long rows) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long w = t >> 6;
    const int lane = (int)(t & 63);
    for (long r0 = w * 4; r0 < rows; r0 += (long)gridDim.x * blockDim.x / 64 * 4)
        if (lane < 48 && r0 + lane / 12 < rows) dst[(r0 + lane / 12) * 24 + lane % 12] = (float)lane;
}

int main() {
    const long bytes = 1L << 30;
    void* buf = nullptr;
    int* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 1, bytes));
    const int blocks = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_b32_kernel, dim3(blocks), dim3(256), 0, 0, (const int*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(read_b128_kernel, dim3(blocks), dim3(256), 0, 0, (const int4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(write_b32_kernel, dim3(blocks), dim3(256), 0, 0, (int*)buf, bytes / 4);
        hipLaunchKernelGGL(write_b128_kernel, dim3(blocks), dim3(256), 0, 0, (int4*)buf, bytes / 16);
        hipLaunchKernelGGL(write_rows12_kernel, dim3(blocks), dim3(256), 0, 0, (float*)buf, bytes / 96);
        CHECK(hipDeviceSynchronize());
    }
    printf("moved %ld bytes per kernel (write_rows12: %ld bytes stored into %ld rows)\n", bytes, bytes / 96 * 48, bytes / 96);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
