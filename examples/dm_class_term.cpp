// One DM class term (distill_baseline.py:344-354) from C, nothing prepared offline: embed the real batch (single-pass f16),
// embed the synthetic clip keeping the pooling decisions (f16 hi+lo pairs), loss = |mean f_real - mean f_syn|^2 and its
// gradient w.r.t. the features (vd_dm_loss), input gradient back to the synthetic pixels (vd_embed_backward), SGD step with
// momentum (vd_sgd_momentum).  Every device buffer is the caller's.
//
//   dm_class_term <dir> <nreal> <T> <H> <W>
//   reads  <dir>/weights.bin (fp32: w0 b0 w1 b1 w2 b2), <dir>/real.bin (nreal,T,3,H,W), <dir>/syn.bin (1,T,3,H,W)
//   writes <dir>/dm_out.bin: loss (1 float), d loss / d syn (T*3*H*W floats), syn after one SGD(lr .5, momentum .5) step
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/vd_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_VD(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s failed with code %d\n", #x, r_); return 3; } } while (0)

static float* read_floats(const char* dir, const char* name, size_t* n) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f) / sizeof(float); fseek(f, 0, SEEK_SET);
    float* p = (float*)malloc(*n * sizeof(float));
    if (fread(p, sizeof(float), *n, f) != *n) { fprintf(stderr, "short read %s\n", path); exit(1); }
    fclose(f);
    return p;
}

static float* to_device(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) exit(2);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s dir nreal T H W\n", argv[0]); return 1; }
    const char* dir = argv[1];
    const int NR = atoi(argv[2]), T = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]);
    const size_t clip = (size_t)T * 3 * H * W;
    VdEmbed *real_net = NULL, *syn_net = NULL;
    CHECK_VD(vd_embed_create(T, H, W, VD_PREC_F16, NR, &real_net));
    CHECK_VD(vd_embed_create_ex(T, H, W, VD_PREC_F16X3, VD_PREC_F16X3, 1, &syn_net));
    const size_t wn[6] = {64u * 3 * 147, 64, 128u * 64 * 147, 128, 128u * 128 * 147, 128};
    size_t n;
    float* hw = read_floats(dir, "weights.bin", &n);
    float* dw[6];
    size_t off = 0;
    for (int i = 0; i < 6; ++i) { dw[i] = to_device(hw + off, wn[i]); off += wn[i]; }
    if (off != n) { fprintf(stderr, "weights.bin has the wrong size\n"); return 1; }
    CHECK_VD(vd_embed_set_weights(real_net, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], NULL));
    CHECK_VD(vd_embed_set_weights(syn_net, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], NULL));
    float* hreal = read_floats(dir, "real.bin", &n);
    if (n != NR * clip) { fprintf(stderr, "real.bin has the wrong size\n"); return 1; }
    float* hsyn = read_floats(dir, "syn.bin", &n);
    if (n != clip) { fprintf(stderr, "syn.bin has the wrong size\n"); return 1; }
    float *dreal = to_device(hreal, NR * clip), *dsyn = to_device(hsyn, clip);
    const int64_t nfeat = vd_embed_num_features(real_net);
    const int64_t ws_real = vd_embed_workspace_bytes(real_net, NR), ws_syn = vd_embed_workspace_bytes(syn_net, 1);
    const int64_t ws_bwd = vd_embed_backward_workspace_bytes(syn_net, 1), am_bytes = vd_embed_argmax_bytes(syn_net, 1);
    int64_t ws_bytes = ws_real > ws_syn ? ws_real : ws_syn;
    if (ws_bwd > ws_bytes) ws_bytes = ws_bwd;
    void* ws; uint8_t* am; float *f_real, *f_syn, *loss, *g_syn, *g_clip, *mom;
    CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
    CHECK_HIP(hipMalloc((void**)&am, (size_t)am_bytes));
    CHECK_HIP(hipMalloc((void**)&f_real, (size_t)NR * nfeat * 4));
    CHECK_HIP(hipMalloc((void**)&f_syn, (size_t)nfeat * 4));
    CHECK_HIP(hipMalloc((void**)&loss, 4));
    CHECK_HIP(hipMalloc((void**)&g_syn, (size_t)nfeat * 4));
    CHECK_HIP(hipMalloc((void**)&g_clip, clip * 4));
    CHECK_HIP(hipMalloc((void**)&mom, clip * 4));
    CHECK_VD(vd_embed_forward(real_net, dreal, NULL, NR, ws, ws_bytes, f_real, NULL));
    CHECK_VD(vd_embed_forward_keep(syn_net, dsyn, NULL, 1, ws, ws_bytes, f_syn, am, NULL));
    CHECK_VD(vd_dm_loss(f_real, f_syn, 1, NR, 1, (int)nfeat, loss, g_syn, NULL));
    CHECK_VD(vd_embed_backward(syn_net, g_syn, am, 1, ws, ws_bytes, g_clip, NULL));
    CHECK_VD(vd_sgd_momentum(dsyn, mom, g_clip, (int64_t)clip, 0.5f, 0.5f, 1, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* out = (float*)malloc((1 + 2 * clip) * sizeof(float));
    CHECK_HIP(hipMemcpy(out, loss, 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + 1, g_clip, clip * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + 1 + clip, dsyn, clip * 4, hipMemcpyDeviceToHost));
    char path[1024];
    snprintf(path, sizeof path, "%s/dm_out.bin", dir);
    FILE* f = fopen(path, "wb"); fwrite(out, sizeof(float), 1 + 2 * clip, f); fclose(f);
    printf("dm_class_term: %d real clips %dx%dx%d, loss %.6f\n", NR, T, H, W, out[0]);
    vd_embed_free(real_net); vd_embed_free(syn_net);
    return 0;
}
