// Probe of the fp8 (OCP e4m3) conversions used by the fp8-corrected hi+lo last level: v_cvt_pk_fp8_f32 and
// v_cvt_scalef32_pk_fp8_f16 (does the scale multiply or divide?  what happens beyond 448?).  Decoding is done on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, float sc, unsigned* o32, unsigned* o16) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float v = in[i];
    o32[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff;
    h2 h = {(_Float16)v, (_Float16)v};
    s2 old = {0, 0};
    s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, h, sc, false);
    o16[i] = (unsigned)(unsigned short)r[0] & 0xff;
}
static double dec(unsigned b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    if (e == 15 && m == 7) return NAN;
    const double v = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1 + m / 8.0, e - 7);
    return s ? -v : v;
}
int main() {
    const float vals[] = {0.f, 1.f, -1.f, 0.3f, 1.0625f, 1.1875f, 17.f, 300.f, 448.f, 460.f, 500.f, 1000.f, 70000.f, 0.01f, 0.002f, 0.001f, -0.37f, 5.5f};
    const int n = sizeof(vals) / sizeof(float);
    float* in; unsigned *a, *b;
    hipMallocManaged(&in, n * 4); hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4);
    for (int i = 0; i < n; ++i) in[i] = vals[i];
    for (float sc : {1.0f, 4.0f, 0.25f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, n, sc, a, b);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        printf("scale operand %g\n", sc);
        for (int i = 0; i < n; ++i)
            printf("  v %10g  cvt_pk_fp8_f32 -> 0x%02x = %-10g   cvt_scalef32_pk_fp8_f16 -> 0x%02x = %-10g\n", vals[i], a[i], dec(a[i]), b[i], dec(b[i]));
    }
    return 0;
}
