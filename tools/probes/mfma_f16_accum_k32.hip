// How v_mfma_f32_16x16x32_f16 accumulates on gfx950: D = C + sum_k a_k b_k over K = 32 products of float16 values
// (each exact in float32).  Every row of A holds the same 32 values a_k and every column of B the same b_k, with a power
// of two p_k = a_k b_k split as a_k = +-2^ceil(e/2), b_k = 2^floor(e/2) so that both stay inside float16's range; all
// 256 results are equal; the printed result is compared with the outcome of candidate models:
//   fused  : the exact sum, rounded once to nearest-even
//   chain  : float32 additions one product at a time, each rounded to nearest-even (an fmaf chain)
//   trunc  : the exact sum, truncated toward zero
// The cases locate the groups of k whose products are summed before they meet C (a tie 1 + 2^-24 at k = 0 is
// rounded up only if a smaller addend at k = kb is still seen in the same addition) and count the guard bits.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Case { float c; float a[32]; };

__global__ void run(const Case* cs, int n, float* out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const float p = cs[t].a[8 * (lane >> 4) + j];
            int e = 0;
            const float m = frexpf(p, &e);                      // p = m 2^e, m = +-0.5 for the powers of two used here
            const int ea = p == 0.0f ? 0 : (e - 1) / 2, eb = p == 0.0f ? 0 : (e - 1) - ea;
            a[j] = (_Float16)(p == 0.0f ? 0.0f : ldexpf(2.0f * m, ea));
            b[j] = (_Float16)ldexpf(1.0f, eb);
        }
        f32x4 c = {cs[t].c, cs[t].c, cs[t].c, cs[t].c};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        float lo = fminf(fminf(c[0], c[1]), fminf(c[2], c[3])), hi = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
        if (lane == 0) { out[2 * t] = lo; out[2 * t + 1] = hi; }
    }
}

static float chain(const Case& k) { float s = k.c; for (int i = 0; i < 32; ++i) s = fmaf(k.a[i], 1.0f, s); return s; }
static double exact(const Case& k) { double s = k.c; for (int i = 0; i < 32; ++i) s += (double)k.a[i]; return s; }
static float trunc32(double x) { float f = (float)x; if (fabs((double)f) > fabs(x)) f = nextafterf(f, 0.0f); return f; }

int main() {
    static Case cs[200];
    static char names[200][72];
    int n = 0;
    auto add = [&](float c) -> Case& { memset(&cs[n], 0, sizeof(Case)); cs[n].c = c; return cs[n++]; };
    const int kbs[] = {1, 7, 8, 9, 15, 16, 17, 23, 24, 31};
    for (int kb : kbs)
        for (int j : {1, 2, 3, 5, 7, 9}) {
            snprintf(names[n], 72, "1 + 2^-24 [k=0] + 2^-%d [k=%d]", 24 + j, kb);
            Case& k = add(1.0f); k.a[0] = ldexpf(1, -24); k.a[kb] = ldexpf(1, -(24 + j));
        }
    for (int kb : {1, 8, 16, 24})          // the tie itself at a later k, the small addend before it
        for (int j : {1, 3, 7}) {
            snprintf(names[n], 72, "1 + 2^-%d [k=0] + 2^-24 [k=%d]", 24 + j, kb);
            Case& k = add(1.0f); k.a[0] = ldexpf(1, -(24 + j)); k.a[kb] = ldexpf(1, -24);
        }
    for (int kb : {1, 8, 16, 24})          // C = 0: the big addend among the products
        for (int sft = 1; sft <= 9; sft += 2) {
            snprintf(names[n], 72, "0 + 1 [k=0] + 2^-24 [k=1] + 3 x 2^-%d [k=%d..]", 23 + sft, kb + 1);
            Case& k = add(0.0f); k.a[0] = 1.0f; k.a[1] = ldexpf(1, -24);
            for (int i = 0; i < 3; ++i) k.a[(kb + 1 + i) % 32] = ldexpf(1, -(23 + sft));
        }
    { snprintf(names[n], 72, "1 + 31 x 2^-25 (7.75 ulp)"); Case& k = add(1.0f); for (int i = 0; i < 31; ++i) k.a[i] = ldexpf(1, -25); }
    { snprintf(names[n], 72, "1 + 32 x 2^-26 (4 ulp)"); Case& k = add(1.0f); for (int i = 0; i < 32; ++i) k.a[i] = ldexpf(1, -26); }
    { snprintf(names[n], 72, "2^24 + 32 x 1"); Case& k = add(ldexpf(1, 24)); for (int i = 0; i < 32; ++i) k.a[i] = 1; }
    { snprintf(names[n], 72, "2^24 + 31 x 1"); Case& k = add(ldexpf(1, 24)); for (int i = 0; i < 31; ++i) k.a[i] = 1; }
    { snprintf(names[n], 72, "0 + 1 [k=0] - 1 [k=16] + 2^-30 [k=24]"); Case& k = add(0.0f); k.a[0] = 1; k.a[16] = -1; k.a[24] = ldexpf(1, -30); }
    { snprintf(names[n], 72, "0 + 1 [k=0] - 1 [k=1] + 2^-30 [k=2]"); Case& k = add(0.0f); k.a[0] = 1; k.a[1] = -1; k.a[2] = ldexpf(1, -30); }
    { snprintf(names[n], 72, "0 + 1 [k=0] - 1 [k=8] + 2^-30 [k=9]"); Case& k = add(0.0f); k.a[0] = 1; k.a[8] = -1; k.a[9] = ldexpf(1, -30); }
    Case* d; float* o; static float h[400];
    if (hipMalloc(&d, sizeof(cs)) != hipSuccess || hipMalloc(&o, sizeof(h)) != hipSuccess) return 1;
    if (hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, d, n, o);
    if (hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("%-52s %-16s %-16s %-16s %-16s\n", "case", "hardware", "fused", "chain", "trunc");
    for (int t = 0; t < n; ++t) {
        const double e = exact(cs[t]);
        printf("%-52s %-16a %-16a %-16a %-16a%s\n", names[t], h[2 * t], (float)e, chain(cs[t]), trunc32(e),
               h[2 * t] != h[2 * t + 1] ? "  (elements differ!)" : "");
    }
    return 0;
}
