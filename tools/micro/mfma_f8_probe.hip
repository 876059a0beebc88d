// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3 operands on gfx950 (the hi+lo corrections of the last
// level at the fp8 rate).  Checks, with exact small-integer data:
//   1. A lane (row r = l & 31, half h = l >> 5) element j and B lane (col c, half h) element j meet in the same k -- i.e. the
//      K map of the two operands is the same function f(h, j), which is all the kernel relies on;
//   2. what the E8M0 scale operands do (D = 2^(sa - 127) * 2^(sb - 127) * sum);
//   3. v_cvt_pk_fp8_f32 produces the e4m3 bytes the MFMA reads;
//   4. the issue rate next to v_mfma_f32_32x32x16_f16 (same output tile, 4x the K).
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_f8_probe tools/micro/mfma_f8_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pk4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// A[r][h][j], B[h][j][c] as floats (small integers / powers of two: exact in e4m3); out D[r][c]
__global__ void probe(const float* A, const float* B, float* D, int sa, int sb) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a, b;
    for (int q = 0; q < 8; ++q) {
        const float* pa = A + (r * 2 + h) * 32 + 4 * q;
        a[q] = (int)pk4_fp8(pa[0], pa[1], pa[2], pa[3]);
        float bv[4];
        for (int e = 0; e < 4; ++e) bv[e] = B[((h * 32) + 4 * q + e) * 32 + r];
        b[q] = (int)pk4_fp8(bv[0], bv[1], bv[2], bv[3]);
    }
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

template <int MODE>
__global__ void rate(float* out, int iters) {
    v16f acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    v8i a, b; v8h ah, bh;
    for (int q = 0; q < 8; ++q) { a[q] = 0x38383838 + threadIdx.x + q; b[q] = 0x3c343c34 ^ (threadIdx.x * 7 + q); ah[q] = (_Float16)(0.01f * (threadIdx.x + q)); bh[q] = (_Float16)(0.02f * q - 0.05f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 0, 0, 0, 127, 0, 127);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *A, *B, *D;
    hipMallocManaged(&A, 32 * 64 * 4); hipMallocManaged(&B, 64 * 32 * 4); hipMallocManaged(&D, 32 * 32 * 4);
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) A[i] = (float)((rand() % 9) - 4);            // [-4, 4]: exact in e4m3
    for (int i = 0; i < 64 * 32; ++i) B[i] = (float)((rand() % 5) - 2) * 0.5f;     // multiples of 0.5
    int bad_total = 0;
    const int scales[3][2] = {{127, 127}, {124, 127}, {127, 130}};
    for (int t = 0; t < 3; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, D, scales[t][0], scales[t][1]);
        hipDeviceSynchronize();
        const double f = ldexp(1.0, scales[t][0] - 127 + scales[t][1] - 127);
        int bad = 0; double maxerr = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)A[r * 64 + k] * B[k * 32 + c];
            const double e = fabs(D[r * 32 + c] - f * ref);
            if (e > 1e-6 * (1 + fabs(ref))) ++bad;
            if (e > maxerr) maxerr = e;
        }
        printf("scales (%d, %d): %d of 1024 outputs differ from 2^%d * sum_k A[r][k] B[k][c] (max err %g; D[0][0] %g)\n", scales[t][0], scales[t][1],
               bad, scales[t][0] - 127 + scales[t][1] - 127, maxerr, D[0]);
        bad_total += bad;
    }
    // fp8 conversion of a few values: print the bytes
    // rate
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 8;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double flop = (double)blocks * 4 * iters * 4 * 32.0 * 32.0 * (mode == 0 ? 16 : 64) * 2.0;
        printf("%s: %.2f ms -> %.0f TFLOP/s\n", mode == 0 ? "v_mfma_f32_32x32x16_f16   " : "v_mfma_scale_32x32x64 fp8 ", best, flop / best / 1e9);
    }
    printf("probe: %s\n", bad_total == 0 ? "ok" : "MISMATCH");
    return bad_total != 0;
}
