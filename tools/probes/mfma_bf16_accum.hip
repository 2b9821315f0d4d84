// How v_mfma_f32_32x32x16_bf16 accumulates on gfx950: D = C + sum_k a_k b_k with products of bf16 values (exact
// in float32).  Each test gives every row of A the same 16 values and every column of B the same 16 values, so
// all 1024 results are equal; the printed result is compared with the outcome of candidate models:
//   fused  : the exact sum, rounded once to nearest-even
//   chain  : float32 additions one product at a time, each rounded to nearest-even (an fmaf chain)
//   trunc  : the exact sum, truncated toward zero
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Case { float c; float a[16]; float b[16]; };

__global__ void run(const Case* cs, int n, float* out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (__bf16)cs[t].a[8 * (lane >> 5) + j];
            b[j] = (__bf16)cs[t].b[8 * (lane >> 5) + j];
        }
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = cs[t].c;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        float lo = c[0], hi = c[0];
        for (int i = 1; i < 16; ++i) { lo = fminf(lo, c[i]); hi = fmaxf(hi, c[i]); }
        if (lane == 0) { out[2 * t] = lo; out[2 * t + 1] = hi; }
    }
}

static float chain(const Case& k) { float s = k.c; for (int i = 0; i < 16; ++i) s = fmaf(k.a[i], k.b[i], s); return s; }
static double exact(const Case& k) { double s = k.c; for (int i = 0; i < 16; ++i) s += (double)k.a[i] * k.b[i]; return s; }
static float trunc32(double x) { float f = (float)x; if (fabs((double)f) > fabs(x)) f = nextafterf(f, 0.0f); return f; }

int main() {
    static Case cs[100];
    const char* names[100];
    int n = 0;
    auto add = [&](const char* name, float c) -> Case& { memset(&cs[n], 0, sizeof(Case)); cs[n].c = c; names[n] = name; for (int i = 0; i < 16; ++i) cs[n].b[i] = 1.0f; return cs[n++]; };
    { Case& k = add("1 + 15 x 2^-25 (3.75 ulp)", 1.0f); for (int i = 0; i < 15; ++i) k.a[i] = ldexpf(1, -25); }
    { Case& k = add("1 + 2 x 2^-24 (1 ulp)", 1.0f); k.a[0] = k.a[1] = ldexpf(1, -24); }
    { Case& k = add("1 + 2^-24 (tie)", 1.0f); k.a[0] = ldexpf(1, -24); }
    { Case& k = add("1 + 2^-23 + 2^-24 (tie, odd)", 1.0f); k.a[0] = ldexpf(1, -23); k.a[1] = ldexpf(1, -24); }
    { Case& k = add("1 + 2^-24 + 2^-40 (above tie)", 1.0f); k.a[0] = ldexpf(1, -24); k.a[5] = ldexpf(1, -40); }
    { Case& k = add("1 + 2^-24 + 2^-60 (above tie, far)", 1.0f); k.a[0] = ldexpf(1, -24); k.a[5] = ldexpf(1, -60); }
    { Case& k = add("0 + 1 - 1 + 2^-30", 0.0f); k.a[0] = 1; k.a[1] = -1; k.a[2] = ldexpf(1, -30); }
    { Case& k = add("0 + 1 - 1 + 2^-60", 0.0f); k.a[0] = 1; k.a[1] = -1; k.a[2] = ldexpf(1, -60); }
    { Case& k = add("0 + 2^20 - 2^20 + 2^-30", 0.0f); k.a[0] = ldexpf(1, 20); k.a[1] = -ldexpf(1, 20); k.a[2] = ldexpf(1, -30); }
    { Case& k = add("2^24 + 16 x 1", ldexpf(1, 24)); for (int i = 0; i < 16; ++i) k.a[i] = 1; }
    { Case& k = add("2^24 + 15 x 1", ldexpf(1, 24)); for (int i = 0; i < 15; ++i) k.a[i] = 1; }
    { Case& k = add("-2^24 + 2^24 + 2^-10 (c cancels)", -ldexpf(1, 24)); k.a[0] = ldexpf(1, 24); k.a[1] = ldexpf(1, -10); }
    { Case& k = add("1 - 2^-25 x 15", 1.0f); for (int i = 0; i < 15; ++i) k.a[i] = -ldexpf(1, -25); }
    for (int s = 22; s <= 34; s += 2) {   // 1 + 2^-23 (odd) + 2^-24 (tie) + 2^-s: where does the sticky bit stop counting?
        static char buf[16][64];
        snprintf(buf[(s - 22) / 2], 64, "1 + 2^-23 + 2^-24 + 2^-%d", s + 2);
        Case& k = add(buf[(s - 22) / 2], 1.0f); k.a[0] = ldexpf(1, -23); k.a[1] = ldexpf(1, -24); k.a[9] = ldexpf(1, -(s + 2));
    }
    // 1 + 2^-24 (tie, at k = 0) + 2^-(24+j) at k = kb: rounds up only while the small addend is still seen
    static char buf[40][64];
    int nb = 0;
    const int kbs[3] = {1, 5, 9};
    for (int q = 0; q < 3; ++q)
        for (int j = 1; j <= 9; j += (j < 4 ? 1 : 2)) {
            snprintf(buf[nb], 64, "1 + 2^-24 [k=0] + 2^-%d [k=%d]", 24 + j, kbs[q]);
            Case& k = add(buf[nb++], 1.0f); k.a[0] = ldexpf(1, -24); k.a[kbs[q]] = ldexpf(1, -(24 + j));
        }
    for (int j = 1; j <= 9; j += 2) {     // the big addend among the products, C = 0
        snprintf(buf[nb], 64, "0 + 1 [k=0] + 2^-24 [k=1] + 2^-%d [k=2]", 24 + j);
        Case& k = add(buf[nb++], 0.0f); k.a[0] = 1.0f; k.a[1] = ldexpf(1, -24); k.a[2] = ldexpf(1, -(24 + j));
    }
    // guard bits of the product sum itself (C = 0): 1 [k=0] + 2^-24 [k=1] (a tie) + m x 2^-(23+s) [k=2..]: rounds up
    // to 1 + ulp only if the addends 2^-(23+s) survive the alignment to the largest product
    for (int sft = 2; sft <= 9; ++sft) {
        snprintf(buf[nb], 64, "0 + 1 + 2^-24 + 3 x 2^-%d", 23 + sft);
        Case& k = add(buf[nb++], 0.0f); k.a[0] = 1.0f; k.a[1] = ldexpf(1, -24);
        for (int i = 2; i < 5; ++i) k.a[i] = ldexpf(1, -(23 + sft));
    }
    { Case& k = add("0 + 1 + 2 x 2^-24", 0.0f); k.a[0] = 1.0f; k.a[1] = k.a[2] = ldexpf(1, -24); }
    { Case& k = add("0 + 1 + 4 x 2^-25", 0.0f); k.a[0] = 1.0f; for (int i = 1; i < 5; ++i) k.a[i] = ldexpf(1, -25); }
    { Case& k = add("0 - 1 - 2^-24 - 3 x 2^-26", 0.0f); k.a[0] = -1.0f; k.a[1] = -ldexpf(1, -24); for (int i = 2; i < 5; ++i) k.a[i] = -ldexpf(1, -26); }
    { Case& k = add("0 + 1 - 2^-25 (just below 1)", 0.0f); k.a[0] = 1.0f; k.a[1] = -ldexpf(1, -25); }
    { Case& k = add("0 + 1 - 2^-26", 0.0f); k.a[0] = 1.0f; k.a[1] = -ldexpf(1, -26); }
    { Case& k = add("0 + 1 - 2^-30", 0.0f); k.a[0] = 1.0f; k.a[1] = -ldexpf(1, -30); }
    Case* d; float* o; static float h[200];
    if (hipMalloc(&d, sizeof(cs)) != hipSuccess || hipMalloc(&o, sizeof(h)) != hipSuccess) return 1;
    if (hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, d, n, o);
    if (hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("%-40s %-16s %-16s %-16s %-16s\n", "case", "hardware", "fused", "chain", "trunc");
    for (int t = 0; t < n; ++t) {
        const double e = exact(cs[t]);
        printf("%-40s %-16a %-16a %-16a %-16a%s\n", names[t], h[2 * t], (float)e, chain(cs[t]), trunc32(e),
               h[2 * t] != h[2 * t + 1] ? "  (elements differ!)" : "");
    }
    return 0;
}
