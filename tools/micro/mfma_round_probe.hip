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

static float rne(long double v) { return (float)v; }      // long double -> float conversion rounds to nearest even
static float trunc0(long double v) {
    float f = (float)v;
    if ((long double)f != v && fabsl((long double)f) > fabsl(v)) f = nextafterf(f, 0.f);
    return f;
}

int main() {
    const int waves = 4096, nsteps = 600;
    std::vector<_Float16> a((size_t)waves * nsteps * 16), b(a.size());
    srand(12345);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (size_t i = 0; i < a.size(); ++i) { a[i] = (_Float16)(rnd() * 1.0f); b[i] = (_Float16)(rnd() * 0.05f); }
    _Float16 *da, *db; float* dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dout, waves * 2 * sizeof(float));
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, da, db, nsteps, dout);
    std::vector<float> out(waves * 2);
    if (hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    int same_elem = 0, m_rne = 0, m_trunc = 0, m_seq = 0;
    long double bias = 0, abs_err = 0, bias_rne = 0;
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
        const long double ulp = ldexpl(1.0L, ilogbl(fabsl(exact)) - 23);
        bias += ((long double)d - exact) / ulp * (exact > 0 ? 1 : -1);
        bias_rne += ((long double)acc_r - exact) / ulp * (exact > 0 ? 1 : -1);
        abs_err += fabsl((long double)d - exact) / ulp;
    }
    printf("v_mfma_f32_32x32x16_f16, %d chains of %d instructions: two elements of a tile equal in %d; device bits == per-instruction RNE model in %d, == truncation "
           "model in %d, == per-product sequential RNE in %d\n", waves, nsteps, same_elem, m_rne, m_trunc, m_seq);
    printf("device vs exact total: mean signed error toward +|result| %.3Lf ulp (RNE model: %.3Lf), mean |error| %.3Lf ulp of the result\n",
           bias / waves, bias_rne / waves, abs_err / waves);
    return 0;
}
