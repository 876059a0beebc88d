// ConvNet3D.embed (networks.py:747-751 of the reference) from C with nothing prepared offline: the library plans the
// tile programs itself (vd_embed_create -> csrc/planner.cpp), the caller owns every device buffer.
//
//   embed_standalone <dir> <nclips> <T> <H> <W> <prec 0..3>
//   reads  <dir>/weights.bin (fp32: w0 b0 w1 b1 w2 b2), <dir>/clips.bin (fp32 B,T,3,H,W);  writes <dir>/feats_standalone.bin
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

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s dir nclips T H W prec\n", argv[0]); return 1; }
    const char* dir = argv[1];
    const int B = atoi(argv[2]), T = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), prec = atoi(argv[6]);
    if (vd_abi_version() != VD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    VdEmbed* net = NULL;
    CHECK_VD(vd_embed_create(T, H, W, prec, B, &net));
    const size_t wn[6] = {64u * 3 * 147, 64, 128u * 64 * 147, 128, 128u * 128 * 147, 128};
    size_t n;
    float* hw = read_floats(dir, "weights.bin", &n);
    float* dw[6];
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
        CHECK_HIP(hipMalloc((void**)&dw[i], wn[i] * sizeof(float)));
        CHECK_HIP(hipMemcpy(dw[i], hw + off, wn[i] * sizeof(float), hipMemcpyHostToDevice));
        off += wn[i];
    }
    if (off != n) { fprintf(stderr, "weights.bin has the wrong size\n"); return 1; }
    CHECK_VD(vd_embed_set_weights(net, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], NULL));
    float* hx = read_floats(dir, "clips.bin", &n);
    if (n != (size_t)B * T * 3 * H * W) { fprintf(stderr, "clips.bin has the wrong size\n"); return 1; }
    float *dx, *feats;
    void* ws;
    const int64_t nfeat = vd_embed_num_features(net), ws_bytes = vd_embed_workspace_bytes(net, B);
    CHECK_HIP(hipMalloc((void**)&dx, n * sizeof(float)));
    CHECK_HIP(hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
    CHECK_HIP(hipMalloc((void**)&feats, (size_t)B * nfeat * sizeof(float)));
    if (vd_embed_forward(net, dx, NULL, B, ws, ws_bytes - 1, feats, NULL) != -7) { fprintf(stderr, "short workspace accepted\n"); return 1; }
    CHECK_VD(vd_embed_forward(net, dx, NULL, B, ws, ws_bytes, feats, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* hf = (float*)malloc((size_t)B * nfeat * sizeof(float));
    CHECK_HIP(hipMemcpy(hf, feats, (size_t)B * nfeat * sizeof(float), hipMemcpyDeviceToHost));
    char path[1024];
    snprintf(path, sizeof path, "%s/feats_standalone.bin", dir);
    FILE* f = fopen(path, "wb"); fwrite(hf, sizeof(float), (size_t)B * nfeat, f); fclose(f);
    printf("embed_standalone: %d clips %dx%dx%d -> %lld features each, workspace %.1f MB\n", B, T, H, W, (long long)nfeat, ws_bytes / 1e6);
    vd_embed_free(net);
    return 0;
}
