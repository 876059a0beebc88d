// How does v_mfma_f32_32x32x16_f16 round?  Every output element of a chain of N MFMAs is made the SAME dot-product chain
// (all rows of A and all columns of B equal per step), so one float per wave comes back; the host emulates three models from the
// same fp16 values in long double -- (a) round-to-nearest-even of (acc + exact 16-term sum) per instruction, (b) truncation toward
// zero of the same, (c) per-product sequential fp32 RNE adds -- and counts which one reproduces the device bits, plus the mean SIGNED
// error of the device result against the exact total (a bias toward zero shows as a negative mean of err * sign(result)).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_round_probe.hip -o tools/micro/mfma_round_probe && tools/micro/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const _Float16* __restrict__ a, const _Float16* __restrict__ b, int nsteps, float* __restrict__ out) {
    const int wave = blockIdx.x, lane = threadIdx.x, half = lane >> 5;
    f32x16 acc;
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int s = 0; s < nsteps; ++s) {
        const _Float16* as = a + ((size_t)wave * nsteps + s) * 16 + half * 8;
        const _Float16* bs = b + ((size_t)wave * nsteps + s) * 16 + half * 8;
        f16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = as[j]; bv[j] = bs[j]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    }
    if (lane == 0) out[wave] = acc[0];
    if (lane == 37) out[gridDim.x + wave] = acc[9];      // another element: must be the same value
}

__global__ void probe_one(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x, half = lane >> 5;
    f32x16 acc;
    for (int k = 0; k < 16; ++k) acc[k] = 16.f;
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[half * 8 + j]; bv[j] = b[half * 8 + j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float rne(long double v) { return (float)v; }      // long double -> float conversion rounds to nearest even
static float trunc0(long double v) {
    float f = (float)v;
    if ((long double)f != v && fabsl((long double)f) > fabsl(v)) f = nextafterf(f, 0.f);
    return f;
}

static void chains(const char* label, int sign_mode) {      // 0: random signs, 1: all products positive, 2: all products negative
    const int waves = 4096, nsteps = 600;
    std::vector<_Float16> a((size_t)waves * nsteps * 16), b(a.size());
    srand(12345);
    auto rnd = [&] { const float u = (float)rand() / RAND_MAX; return sign_mode ? u : u * 2.f - 1.f; };
    for (size_t i = 0; i < a.size(); ++i) { a[i] = (_Float16)(rnd() * 1.0f); b[i] = (_Float16)(rnd() * (sign_mode == 2 ? -0.05f : 0.05f)); }
    _Float16 *da, *db; float* dout;
    (void)hipMalloc(&da, a.size() * 2); (void)hipMalloc(&db, b.size() * 2); (void)hipMalloc(&dout, waves * 2 * sizeof(float));
    (void)hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, da, db, nsteps, dout);
    std::vector<float> out(waves * 2);
    if (hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    int same_elem = 0, m_rne = 0, m_trunc = 0, m_seq = 0;
    long double bias = 0, abs_err = 0, bias_rne = 0, abs_rne = 0, bias_tr = 0, plain = 0, plain_rne = 0;
    for (int w = 0; w < waves; ++w) {
        float acc_r = 0.f, acc_t = 0.f, acc_s = 0.f;
        long double exact = 0;
        for (int s = 0; s < nsteps; ++s) {
            long double dot = 0;
            for (int k = 0; k < 16; ++k) {
                const long double p = (long double)(float)a[((size_t)w * nsteps + s) * 16 + k] * (long double)(float)b[((size_t)w * nsteps + s) * 16 + k];
                dot += p;
                acc_s = rne((long double)acc_s + p);
            }
            exact += dot;
            acc_r = rne((long double)acc_r + dot);
            acc_t = trunc0((long double)acc_t + dot);
        }
        const float d = out[w];
        same_elem += (out[w] == out[waves + w]);
        m_rne += (d == acc_r); m_trunc += (d == acc_t); m_seq += (d == acc_s);
        const long double ulp = ldexpl(1.0L, ilogbl(fabsl(exact)) - 23), sg = exact > 0 ? 1 : -1;
        bias += ((long double)d - exact) / ulp * sg;
        bias_rne += ((long double)acc_r - exact) / ulp * sg;
        bias_tr += ((long double)acc_t - exact) / ulp * sg;
        plain += ((long double)d - exact) / ulp;
        plain_rne += ((long double)acc_r - exact) / ulp;
        abs_err += fabsl((long double)d - exact) / ulp;
        abs_rne += fabsl((long double)acc_r - exact) / ulp;
    }
    printf("%s: %d chains of %d v_mfma_f32_32x32x16_f16 (two elements of a tile equal in %d): device bits == per-instruction RNE model in %d, == per-instruction "
           "truncation model in %d, == per-product sequential RNE in %d\n", label, waves, nsteps, same_elem, m_rne, m_trunc, m_seq);
    printf("   error vs the exact total in ulps of the result, signed toward larger magnitude: device mean %+.2Lf (mean |.| %.2Lf); RNE model %+.2Lf (%.2Lf); truncation model %+.2Lf\n",
           bias / waves, abs_err / waves, bias_rne / waves, abs_rne / waves, bias_tr / waves);
    printf("   plain signed error (device - exact, positive = toward +infinity): device mean %+.2Lf ulp; RNE model %+.2Lf ulp\n", plain / waves, plain_rne / waves);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
}

// one instruction: acc = 16 (ulp 2^-19) and 16 equal products of 2^-(20+g) each (exact sum 2^-(16+g) = 2^(3-g) ulp), all operands NORMAL
// fp16 numbers: how many bits below the accumulator's ulp take part in the sum?  With the exact sum rounded once the result is
// 16 + 2^(3-g) ulp for g <= 3 and 16 for g = 4 (a tie at half an ulp: to even); products rounded one by one to a few guard bits
// below the accumulator's ulp vanish earlier.  MEASURED (profiles/r06_mfma_rounding.txt): g <= 2 exact, g = 3 (products of 1/16
// ulp, sum 1 ulp) gives +0: every product is rounded to 1/8 ulp of the accumulator -- three guard bits -- before the sum; the chains
// above show that this rounding is to nearest (no drift toward zero) and costs 1.4x the error of the ideal per-instruction model.
static void guard_bits() {
    printf("one instruction, acc = 16, 16 products of 2^-(20+g) (ulp of the accumulator 2^-19):");
    for (int g = 0; g <= 4; ++g) {
        std::vector<_Float16> a(16), b(16);
        for (int k = 0; k < 16; ++k) { a[k] = (_Float16)ldexpf(1.f, -10); b[k] = (_Float16)ldexpf(1.f, -(10 + g)); }
        _Float16 *da, *db; float* dout;
        (void)hipMalloc(&da, 32); (void)hipMalloc(&db, 32); (void)hipMalloc(&dout, 8);
        (void)hipMemcpy(da, a.data(), 32, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_one, dim3(1), dim3(64), 0, 0, da, db, dout);
        float out[2];
        (void)hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost);
        printf("  g=%d: 16 + %.3g ulp (exact sum %.3g ulp)", g, (out[0] - 16.f) / ldexpf(1.f, -19), ldexpf(1.f, -(16 + g)) / ldexpf(1.f, -19));
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    }
    printf("\n");
}

int main() {
    chains("random signs", 0);
    chains("all positive", 1);
    chains("all negative", 2);
    guard_bits();
    return 0;
}
