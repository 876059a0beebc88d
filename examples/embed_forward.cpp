// Torch-free, Python-free driver of the hot path through the C ABI (include/vd_hip.h):
// ConvNet3D.embed (networks.py:747-751 of the reference) for a batch of clips, from three serialised
// tile programs written by tools/export_programs.py.
//
//   embed_forward <dir> <nclips> <T> <H> <W> <prec 0..3>
//   reads  <dir>/fwd{0,1,2}.vdprog, <dir>/weights.bin (fp32: w0 b0 w1 b1 w2 b2), <dir>/clips.bin (fp32 B,T,3,H,W)
//   writes <dir>/feats.bin (fp32 B x features)
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/vd_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_VD(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s failed with code %d\n", #x, r_); return 3; } } while (0)

static void* read_file(const char* dir, const char* name, size_t* n) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    void* p = malloc(*n);
    if (fread(p, 1, *n, f) != *n) { fprintf(stderr, "short read %s\n", path); exit(1); }
    fclose(f);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s dir nclips T H W prec\n", argv[0]); return 1; }
    const char* dir = argv[1];
    const int B = atoi(argv[2]), T = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), prec = atoi(argv[6]);
    const int planes = (prec >= 2) ? 2 : 1;
    if (vd_abi_version() != VD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    // programs
    VdProgram* prog[3];
    for (int l = 0; l < 3; ++l) {
        char name[32]; size_t n;
        snprintf(name, sizeof name, "fwd%d.vdprog", l);
        void* blob = read_file(dir, name, &n);
        CHECK_VD(vd_program_load(blob, (int64_t)n, prec, &prog[l]));
        free(blob);
    }
    // weights: (64,3,3,7,7) (64) (128,64,3,7,7) (128) (128,128,3,7,7) (128)
    const size_t wn[6] = {64u * 3 * 147, 64, 128u * 64 * 147, 128, 128u * 128 * 147, 128};
    size_t nb; float* hw = (float*)read_file(dir, "weights.bin", &nb);
    float* dw[6]; size_t off = 0;
    for (int i = 0; i < 6; ++i) {
        CHECK_HIP(hipMalloc((void**)&dw[i], wn[i] * sizeof(float)));
        CHECK_HIP(hipMemcpy(dw[i], hw + off, wn[i] * sizeof(float), hipMemcpyHostToDevice));
        off += wn[i];
    }
    if (off * sizeof(float) != nb) { fprintf(stderr, "weights.bin has the wrong size\n"); return 1; }
    for (int l = 0; l < 3; ++l) CHECK_VD(vd_program_pack_weights(prog[l], dw[2 * l], NULL));

    // clips -> 16-bit pixel rows
    float* hx = (float*)read_file(dir, "clips.bin", &nb);
    const size_t clip_elems = (size_t)T * 3 * H * W;
    if (nb != (size_t)B * clip_elems * sizeof(float)) { fprintf(stderr, "clips.bin has the wrong size\n"); return 1; }
    float* dx; CHECK_HIP(hipMalloc((void**)&dx, nb)); CHECK_HIP(hipMemcpy(dx, hx, nb, hipMemcpyHostToDevice));
    const int rowp = ((W + 8 + 7) / 8) * 8;
    const int64_t slots0 = (int64_t)B * T * 3 * H * (rowp / 8);
    char* rows; CHECK_HIP(hipMalloc((void**)&rows, (size_t)planes * slots0 * 16));
    CHECK_VD(vd_pix2rows(dx, NULL, B, T, H, W, rows, planes == 2 ? rows + slots0 * 16 : NULL, prec, NULL));

    // layer 0 / 1: channels-last 16-bit slots; layer 2: fp32 features
    const int64_t n1 = (int64_t)B * vd_program_info(prog[0], 3), n2 = (int64_t)B * vd_program_info(prog[1], 3);
    const int64_t nfeat = vd_program_info(prog[2], 3);
    char *act1, *act2; float* feats;
    CHECK_HIP(hipMalloc((void**)&act1, (size_t)planes * n1 * 16));
    CHECK_HIP(hipMalloc((void**)&act2, (size_t)planes * n2 * 16));
    CHECK_HIP(hipMalloc((void**)&feats, (size_t)B * nfeat * sizeof(float)));
    CHECK_VD(vd_program_run(prog[0], rows, slots0, dw[1], act1, n1, NULL, NULL, B, NULL));
    CHECK_VD(vd_program_run(prog[1], act1, n1, dw[3], act2, n2, NULL, NULL, B, NULL));
    CHECK_VD(vd_program_run(prog[2], act2, n2, dw[5], feats, 0, NULL, NULL, B, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float* hf = (float*)malloc((size_t)B * nfeat * sizeof(float));
    CHECK_HIP(hipMemcpy(hf, feats, (size_t)B * nfeat * sizeof(float), hipMemcpyDeviceToHost));
    char path[1024]; snprintf(path, sizeof path, "%s/feats.bin", dir);
    FILE* f = fopen(path, "wb"); fwrite(hf, sizeof(float), (size_t)B * nfeat, f); fclose(f);
    printf("embed_forward: %d clips %dx%dx%d -> %lld features each\n", B, T, H, W, (long long)nfeat);
    for (int l = 0; l < 3; ++l) vd_program_free(prog[l]);
    return 0;
}
