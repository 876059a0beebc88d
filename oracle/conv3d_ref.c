/* Plain-C restatement of one ConvNet3D level (TEST INFRASTRUCTURE ONLY -- see oracle/ref_cpu.py).
 *
 * Conv3d(k 3x7x7, stride 1x2x2, pad 1x3x3) -> ReLU -> MaxPool3d(pt x 2 x 2, floor) exactly as the
 * reference composes them (networks.py:768-770, 799), accumulated in double so that it is an
 * arbitration reference independent of any BLAS/MKL-DNN summation order, plus the matching
 * input gradient.  Layout: x [C][T][H][W], w [N][C][3][7][7], y [N][To][Ho][Wo].
 * Pinned against torch (and through it against the reference's golden vectors) in
 * tests/test_oracle_c.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KT 3
#define KH 7
#define KW 7

static int out_dim(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

/* conv grid (pre-activation), double accumulate, float output */
void vdref_conv3d(const float* x, const float* w, const float* b, int C, int T, int H, int W, int N, float* y) {
    const int OT = out_dim(T, KT, 1, 1), OH = out_dim(H, KH, 2, 3), OW = out_dim(W, KW, 2, 3);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < OT; ++t)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    double acc = b ? (double)b[n] : 0.0;
                    for (int c = 0; c < C; ++c)
                        for (int kt = 0; kt < KT; ++kt) {
                            const int it = t + kt - 1;
                            if (it < 0 || it >= T) continue;
                            for (int kh = 0; kh < KH; ++kh) {
                                const int ih = 2 * oh + kh - 3;
                                if (ih < 0 || ih >= H) continue;
                                for (int kw = 0; kw < KW; ++kw) {
                                    const int iw = 2 * ow + kw - 3;
                                    if (iw < 0 || iw >= W) continue;
                                    acc += (double)x[((size_t)(c * T + it) * H + ih) * W + iw] *
                                           (double)w[((((size_t)n * C + c) * KT + kt) * KH + kh) * KW + kw];
                                }
                            }
                        }
                    y[((size_t)(n * OT + t) * OH + oh) * OW + ow] = (float)acc;
                }
}

/* ReLU + MaxPool3d(pt,2,2); arg = window index dt*4+dh*2+dw (dh*2+dw when pt==1) of the FIRST maximum */
void vdref_relu_maxpool(const float* y, int N, int OT, int OH, int OW, int pt, float* p, uint8_t* arg) {
    const int To = OT / pt, Ho = OH / 2, Wo = OW / 2;
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < To; ++t)
            for (int h = 0; h < Ho; ++h)
                for (int wv = 0; wv < Wo; ++wv) {
                    float best = 0.f; int bj = -1;
                    for (int dt = 0; dt < pt; ++dt)
                        for (int dh = 0; dh < 2; ++dh)
                            for (int dw = 0; dw < 2; ++dw) {
                                float v = y[((size_t)(n * OT + t * pt + dt) * OH + 2 * h + dh) * OW + 2 * wv + dw];
                                v = v > 0.f ? v : 0.f;
                                const int j = (pt == 2 ? dt * 4 : 0) + dh * 2 + dw;
                                if (bj < 0 || v > best) { best = v; bj = j; }
                            }
                    const size_t o = ((size_t)(n * To + t) * Ho + h) * Wo + wv;
                    p[o] = best;
                    if (arg) arg[o] = (uint8_t)bj;
                }
}

/* dx = conv3d^T(dy): input gradient of vdref_conv3d, double accumulate */
void vdref_conv3d_bwd_data(const float* dy, const float* w, int C, int T, int H, int W, int N, float* dx) {
    const int OT = out_dim(T, KT, 1, 1), OH = out_dim(H, KH, 2, 3), OW = out_dim(W, KW, 2, 3);
    double* acc = (double*)calloc((size_t)C * T * H * W, sizeof(double));
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < OT; ++t)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    const double g = dy[((size_t)(n * OT + t) * OH + oh) * OW + ow];
                    if (g == 0.0) continue;
                    for (int c = 0; c < C; ++c)
                        for (int kt = 0; kt < KT; ++kt) {
                            const int it = t + kt - 1;
                            if (it < 0 || it >= T) continue;
                            for (int kh = 0; kh < KH; ++kh) {
                                const int ih = 2 * oh + kh - 3;
                                if (ih < 0 || ih >= H) continue;
                                for (int kw = 0; kw < KW; ++kw) {
                                    const int iw = 2 * ow + kw - 3;
                                    if (iw < 0 || iw >= W) continue;
                                    acc[((size_t)(c * T + it) * H + ih) * W + iw] +=
                                        g * (double)w[((((size_t)n * C + c) * KT + kt) * KH + kh) * KW + kw];
                                }
                            }
                        }
                }
    for (size_t i = 0; i < (size_t)C * T * H * W; ++i) dx[i] = (float)acc[i];
    free(acc);
}
