// Serialised tile programs: the C-only entry to the convolution kernel.
//
// A tile program is geometry-specific data computed offline by the planner (video_distillation_amd/plan.py,
// exported with engine.export_program) -- box tables, gather table, tap offsets, weight gather index.  These
// entry points let a caller WITHOUT Python or torch load such a blob, pack weights and launch the layer:
// they own only the small device-resident tables of the program; activations, weights and outputs stay
// caller-owned device buffers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/vd_hip.h"

struct VdProgram {
    VdConvParams p;
    int32_t* d_tables;     // one allocation: type_desc | tables | boxes | gather | widx | col_off
    const int32_t* d_widx; // the weight gather index inside d_tables
    void* d_zero;
    void* d_wpk;
    int64_t n_widx;
    int planes;
};

static const char VD_PROG_MAGIC[8] = {'V', 'D', 'P', 'R', 'O', 'G', '0', '1'};
enum { H_MAGIC = 0, H_CC, H_S, H_NT, H_MW, H_MTW, H_NTW, H_EPI, H_POOL_T, H_RELU, H_N_OUT, H_N_STRIDE, H_OUT_CLIP, H_OUT_CHUNK,
       H_OUT_T, H_LDS_PLANE, H_NTYPES, H_TAB0, H_TAB1, H_TAB2, H_ATOMIC, H_W_BOX, H_NCL, H_NBOX, H_GSTRIDE, H_SRC_CLIP4,
       H_SRC_CHUNK4, H_MT_VALID, H_N_DESC, H_N_TABLES, H_N_BOXES, H_N_GATHER, H_N_WIDX, H_N_COLOFF, H_PERSIST, H_ROW_PLANES, H_ROW_ROWS, H_PAIR_FLIP, H_WORDS = 40 };

extern "C" int vd_program_load(const void* blob, int64_t nbytes, int prec, VdProgram** out) {
    if (blob == nullptr || out == nullptr || nbytes < (int64_t)(H_WORDS * sizeof(int64_t))) return -1;
    const int64_t* h = reinterpret_cast<const int64_t*>(blob);
    if (memcmp(h, VD_PROG_MAGIC, 8) != 0) return -2;
    const int64_t n_desc = h[H_N_DESC], n_tab = h[H_N_TABLES], n_box = h[H_N_BOXES], n_gat = h[H_N_GATHER], n_w = h[H_N_WIDX],
                  n_col = h[H_N_COLOFF];
    if (n_desc < 0 || n_tab < 0 || n_box < 0 || n_gat < 0 || n_w < 0 || n_col < 0) return -3;      // every section count
    const int64_t n_int = n_desc + n_tab + n_box + n_gat + n_w + n_col;
    if (nbytes != (int64_t)(H_WORDS * sizeof(int64_t)) + n_int * (int64_t)sizeof(int32_t)) return -3;
    if (h[H_NBOX] < 0 || h[H_GSTRIDE] < 0 || n_gat != h[H_NBOX] * h[H_GSTRIDE] || n_box != 8 * h[H_NBOX]) return -3;
    if (prec < 0 || prec > 3) return -4;
    VdProgram* g = static_cast<VdProgram*>(calloc(1, sizeof(VdProgram)));
    if (g == nullptr) return -5;
    const int32_t* host = reinterpret_cast<const int32_t*>(h + H_WORDS);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->d_tables), (size_t)(n_int > 0 ? n_int : 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(g->d_tables, host, (size_t)n_int * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&g->d_zero, 64);
    if (e == hipSuccess) e = hipMemset(g->d_zero, 0, 64);
    g->planes = (prec == VD_PREC_BF16X3 || prec == VD_PREC_F16X3) ? 2 : 1;
    g->n_widx = n_w;
    if (e == hipSuccess && n_w > 0) e = hipMalloc(&g->d_wpk, (size_t)g->planes * n_w * sizeof(uint16_t));
    if (e != hipSuccess) {
        if (g->d_tables) (void)hipFree(g->d_tables);
        if (g->d_zero) (void)hipFree(g->d_zero);
        free(g);
        return (int)e;
    }
    VdConvParams& p = g->p;
    const int32_t* d = g->d_tables;
    p.type_desc = d;            d += n_desc;
    p.tables = d;               d += n_tab;
    p.boxes = d;                d += n_box;
    p.gather = d;               d += n_gat;
    g->d_widx = d;              d += n_w;
    p.col_off = n_col > 0 ? d : nullptr;
    p.gather_stride = h[H_GSTRIDE];
    p.zero_slot = g->d_zero;
    p.nbox = (int)h[H_NBOX]; p.ncl = (int)h[H_NCL];
    p.CC = (int)h[H_CC]; p.S = (int)h[H_S]; p.NT = (int)h[H_NT]; p.MW = (int)h[H_MW]; p.MTW = (int)h[H_MTW]; p.NTW = (int)h[H_NTW];
    p.epi = (int)h[H_EPI]; p.pool_t = (int)h[H_POOL_T]; p.relu = (int)h[H_RELU];
    p.n_out = (int)h[H_N_OUT]; p.n_stride = (int)h[H_N_STRIDE];
    p.out_clip_stride = h[H_OUT_CLIP]; p.out_chunk_stride = (int)h[H_OUT_CHUNK]; p.out_t_stride = (int)h[H_OUT_T];
    p.lds_plane_bytes = (int)h[H_LDS_PLANE]; p.ntypes = (int)h[H_NTYPES];
    p.tab_ofs[0] = (int)h[H_TAB0]; p.tab_ofs[1] = (int)h[H_TAB1]; p.tab_ofs[2] = (int)h[H_TAB2];
    p.atomic = (int)h[H_ATOMIC]; p.w_box_stride = h[H_W_BOX];
    p.src_clip_stride4 = h[H_SRC_CLIP4]; p.src_chunk_stride4 = h[H_SRC_CHUNK4];
    p.mt_valid = (int)h[H_MT_VALID]; p.persist = (int)h[H_PERSIST];
    p.src_planes = (int)h[H_ROW_PLANES]; p.src_rows = (int)h[H_ROW_ROWS];     // > 0: the patch can be built from aligned row loads
    p.pair_flip = (int)h[H_PAIR_FLIP];                                        // != 0: frame-tile program (vd_hip.h)
    p.prec = prec;
    p.wpk = g->d_wpk; p.w_plane_stride = n_w;
    *out = g;
    return 0;
}

extern "C" int vd_program_pack_weights(VdProgram* g, const float* w, void* stream) {
    if (g == nullptr || w == nullptr || g->n_widx <= 0) return -1;
    const int32_t* widx = g->d_widx;
    uint16_t* hi = static_cast<uint16_t*>(g->d_wpk);
    return vd_pack_weights(w, widx, g->n_widx, hi, g->planes == 2 ? hi + g->n_widx : nullptr, g->p.prec, stream);
}

extern "C" int vd_program_run(VdProgram* g, const void* src, int64_t src_plane_slots, const float* bias, void* dst,
                              int64_t dst_plane_stride, uint8_t* argmax, const int64_t* clip_index, int nclips, void* stream) {
    return vd_program_run_scaled(g, src, src_plane_slots, bias, dst, dst_plane_stride, argmax, clip_index, nclips, nullptr, stream);
}

extern "C" int vd_program_run_scaled(VdProgram* g, const void* src, int64_t src_plane_slots, const float* bias, void* dst,
                                     int64_t dst_plane_stride, uint8_t* argmax, const int64_t* clip_index, int nclips,
                                     const float* out_scale, void* stream) {
    if (g == nullptr || src == nullptr || dst == nullptr || nclips < 0) return -1;
    VdConvParams p = g->p;         // per-call copy: the handle may be used from several streams
    p.src = src; p.src_plane_stride4 = src_plane_slots * 4;
    p.bias = bias; p.dst = dst; p.dst_plane_stride = dst_plane_stride;
    p.argmax = argmax; p.clip_index = clip_index; p.nclips = nclips; p.out_scale = out_scale;
    // first-layer programs in the single-pass formats and the 2 x 2-wave layout run the kernel that keeps the layer's B
    // fragments in registers across its box walk (bitwise the same results; same rule as engine._DevPlan)
    const bool breg = p.prec < 2 && p.epi == VD_EPI_POOL_CL && p.pool_t == 1 && p.CC == 1 && p.ncl == 1 && p.NTW <= 1 &&
                      p.NT == 2 && p.MW == 2 && p.MTW == 4 && p.S == 32 && p.ntypes == 1 && p.src_rows > 0 &&
                      p.relu && argmax == nullptr;
    return breg ? vd_conv0_breg(&p, stream) : vd_conv_mfma(&p, stream);
}

// A weight-gradient program (vd_program_build_wgrad): x re-laid out clip-minor (vd_clip_minor_cl / _pix) is the source, the
// packed dy of the program's blocks (vd_pack_dy / vd_unpool_relu_bwd_packed) the per-call B operand, and the boxes ACCUMULATE
// (fp32 atomics) into `copies` = [replicas][cin*147][cout] floats, zeroed by the caller and folded into dW by vd_replica_sum.
extern "C" int vd_program_run_wgrad(VdProgram* g, const void* x_clip_minor, int64_t x_plane_slots, const void* packed_dy,
                                    int64_t packed_plane_elems, float* copies, int64_t copy_elems, int cin, const float* out_scale,
                                    void* stream) {
    if (g == nullptr || x_clip_minor == nullptr || packed_dy == nullptr || copies == nullptr || cin <= 0) return -1;
    if (!g->p.atomic || g->p.w_box_stride == 0 || g->p.epi != VD_EPI_ROWS) return -2;      // not a weight-gradient program
    if (copy_elems != (int64_t)cin * g->p.out_clip_stride || copy_elems > 0x7fffffff) return -2;
    VdConvParams p = g->p;
    p.src = x_clip_minor; p.src_plane_stride4 = x_plane_slots * 4;
    p.wpk = packed_dy; p.w_plane_stride = packed_plane_elems;
    p.bias = nullptr; p.dst = copies; p.dst_plane_stride = 0; p.replica_stride = (int32_t)copy_elems;
    p.argmax = nullptr; p.clip_index = nullptr; p.nclips = cin; p.out_scale = out_scale;
    return vd_conv_mfma(&p, stream);
}

extern "C" int64_t vd_program_info(const VdProgram* g, int what) {
    if (g == nullptr) return -1;
    switch (what) {
        case 0: return g->p.nbox;
        case 1: return g->n_widx;
        case 2: return g->planes;
        case 3: return g->p.out_clip_stride;
        case 4: return g->p.n_out;
        default: return -1;
    }
}

extern "C" void vd_program_free(VdProgram* g) {
    if (g == nullptr) return;
    if (g->d_tables) (void)hipFree(g->d_tables);
    if (g->d_zero) (void)hipFree(g->d_zero);
    if (g->d_wpk) (void)hipFree(g->d_wpk);
    free(g);
}


// ---- ConvNet3D.embed as ONE handle: planner (csrc/planner.cpp) + programs + launches -----------------------------------
struct VdEmbed {
    VdProgram* prog[3];
    VdProgram* bwd[3][4];            // input-gradient programs per level (level 0: one merged program)
    int nbwd[3];
    int frames, height, width, prec, planes, prec_bwd, planes_bwd;
    int dims[3][12];                 // cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pool_t per level (plan.NetGeometry.layer_dims)
    int64_t slots0_per_clip, slots1_per_clip, slots2_per_clip, nfeat;
    const float* bias[3];
    bool has_weights;
};

static int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

extern "C" int vd_embed_create_ex(int frames, int height, int width, int prec, int prec_bwd, int batch_hint, VdEmbed** out) {
    if (out == nullptr) return -1;
    VdEmbed* e = static_cast<VdEmbed*>(calloc(1, sizeof(VdEmbed)));
    if (e == nullptr) return -5;
    for (int l = 0; l < 3; ++l) {
        void* blob = nullptr;
        int64_t n = 0;
        int rc = vd_program_build(l, frames, height, width, prec, batch_hint, &blob, &n);
        if (rc == 0) { rc = vd_program_load(blob, n, prec, &e->prog[l]); vd_blob_free(blob); }
        if (rc != 0) { vd_embed_free(e); return rc; }
    }
    e->frames = frames; e->height = height; e->width = width; e->prec = prec;
    e->planes = (prec == VD_PREC_BF16X3 || prec == VD_PREC_F16X3) ? 2 : 1;
    e->prec_bwd = prec_bwd;
    e->planes_bwd = (prec_bwd == VD_PREC_BF16X3 || prec_bwd == VD_PREC_F16X3) ? 2 : 1;
    {
        const int widths[3] = {64, 128, 128}, pools[3] = {1, 2, 2};
        int cin = 3, t = frames, h = height, w = width;
        for (int l = 0; l < 3; ++l) {
            const int T = conv_out(t, 3, 1, 1), OH = conv_out(h, 7, 2, 3), OW = conv_out(w, 7, 2, 3);
            const int d[12] = {cin, widths[l], t, h, w, T, OH, OW, T / pools[l], OH / 2, OW / 2, pools[l]};
            memcpy(e->dims[l], d, sizeof(d));
            cin = widths[l]; t = T / pools[l]; h = OH / 2; w = OW / 2;
        }
    }
    if (prec_bwd >= 0) {
        if (prec_bwd > 3) { vd_embed_free(e); return -1; }
        for (int l = 0; l < 3; ++l) {
            const int ncls = (l == 0) ? 1 : (e->dims[l][3] < 2 ? 1 : 2) * (e->dims[l][4] < 2 ? 1 : 2);
            for (int c = 0; c < ncls; ++c) {
                void* blob = nullptr;
                int64_t n = 0;
                int rc = vd_program_build_dgrad(l, c, frames, height, width, batch_hint, &blob, &n);
                if (rc == 0) { rc = vd_program_load(blob, n, prec_bwd, &e->bwd[l][c]); vd_blob_free(blob); }
                if (rc != 0) { vd_embed_free(e); return rc; }
                e->nbwd[l] = c + 1;
            }
        }
    }
    const int rowp = ((width + 8 + 7) / 8) * 8;
    e->slots0_per_clip = (int64_t)frames * 3 * height * (rowp / 8);
    e->slots1_per_clip = e->prog[0]->p.out_clip_stride;
    e->slots2_per_clip = e->prog[1]->p.out_clip_stride;
    e->nfeat = e->prog[2]->p.out_clip_stride;
    *out = e;
    return 0;
}

extern "C" int vd_embed_create(int frames, int height, int width, int prec, int batch_hint, VdEmbed** out) {
    return vd_embed_create_ex(frames, height, width, prec, -1, batch_hint, out);
}

extern "C" int64_t vd_embed_num_features(const VdEmbed* e) { return e ? e->nfeat : -1; }

extern "C" int64_t vd_embed_workspace_bytes(const VdEmbed* e, int64_t nclips) {
    if (e == nullptr || nclips < 0) return -1;
    return (int64_t)e->planes * nclips * (e->slots0_per_clip + e->slots1_per_clip + e->slots2_per_clip) * 16 + 3 * 256;
}

extern "C" int64_t vd_embed_argmax_bytes(const VdEmbed* e, int64_t nclips) {
    if (e == nullptr || nclips < 0) return -1;
    return nclips * (e->slots1_per_clip * 8 + e->slots2_per_clip * 8 + e->nfeat) + 3 * 256;
}

static int64_t dy_slots(const VdEmbed* e, int l, int64_t nclips) {
    const int* d = e->dims[l];
    return nclips * (d[1] / 8) * (int64_t)d[5] * d[6] * d[7];
}

extern "C" int64_t vd_embed_backward_workspace_bytes(const VdEmbed* e, int64_t nclips) {
    if (e == nullptr || nclips < 0 || e->prec_bwd < 0) return -1;
    int64_t dy = 0;
    for (int l = 0; l < 3; ++l) dy = dy > dy_slots(e, l, nclips) ? dy : dy_slots(e, l, nclips);
    int64_t dx = 0;
    for (int l = 1; l < 3; ++l) dx += nclips * (int64_t)e->dims[l][2] * e->dims[l][3] * e->dims[l][4] * e->dims[l][0] * 4;
    return (int64_t)e->planes_bwd * dy * 16 + dx + 5 * 256;       // four 256-byte alignments + the 12 scale floats behind the last one
}

extern "C" int vd_embed_set_weights(VdEmbed* e, const float* w0, const float* b0, const float* w1, const float* b1,
                                    const float* w2, const float* b2, void* stream) {
    if (e == nullptr || !w0 || !b0 || !w1 || !b1 || !w2 || !b2) return -1;
    const float* w[3] = {w0, w1, w2};
    for (int l = 0; l < 3; ++l) {
        int rc = vd_program_pack_weights(e->prog[l], w[l], stream);
        for (int c = 0; rc == 0 && c < e->nbwd[l]; ++c) rc = vd_program_pack_weights(e->bwd[l][c], w[l], stream);
        if (rc != 0) return rc;
    }
    e->bias[0] = b0; e->bias[1] = b1; e->bias[2] = b2;       // read by the launches: the caller keeps them alive
    e->has_weights = true;
    return 0;
}

static char* align256(char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255); }

static int embed_forward_impl(VdEmbed* e, const float* clips, const int64_t* clip_index, int64_t nclips, void* workspace,
                              int64_t workspace_bytes, float* features, uint8_t* argmax, void* stream) {
    if (e == nullptr || clips == nullptr || features == nullptr || nclips < 0 || nclips > 0x7fffffff) return -1;
    if (!e->has_weights) return -6;
    if (nclips == 0) return 0;
    if (workspace == nullptr || workspace_bytes < vd_embed_workspace_bytes(e, nclips)) return -7;
    const int64_t n0 = nclips * e->slots0_per_clip, n1 = nclips * e->slots1_per_clip, n2 = nclips * e->slots2_per_clip;
    char* rows = align256(static_cast<char*>(workspace));
    char* act1 = align256(rows + (int64_t)e->planes * n0 * 16);
    char* act2 = align256(act1 + (int64_t)e->planes * n1 * 16);
    uint8_t *am0 = nullptr, *am1 = nullptr, *am2 = nullptr;
    if (argmax != nullptr) {
        am0 = reinterpret_cast<uint8_t*>(align256(reinterpret_cast<char*>(argmax)));
        am1 = reinterpret_cast<uint8_t*>(align256(reinterpret_cast<char*>(am0) + n1 * 8));
        am2 = reinterpret_cast<uint8_t*>(align256(reinterpret_cast<char*>(am1) + n2 * 8));
    }
    int rc = vd_pix2rows(clips, clip_index, nclips, e->frames, e->height, e->width, rows, e->planes == 2 ? rows + n0 * 16 : nullptr,
                         e->prec, stream);
    if (rc == 0) rc = vd_program_run(e->prog[0], rows, n0, e->bias[0], act1, n1, am0, nullptr, (int)nclips, stream);
    if (rc == 0) rc = vd_program_run(e->prog[1], act1, n1, e->bias[1], act2, n2, am1, nullptr, (int)nclips, stream);
    if (rc == 0) rc = vd_program_run(e->prog[2], act2, n2, e->bias[2], features, 0, am2, nullptr, (int)nclips, stream);
    return rc;
}

extern "C" int vd_embed_forward(VdEmbed* e, const float* clips, const int64_t* clip_index, int64_t nclips, void* workspace,
                                int64_t workspace_bytes, float* features, void* stream) {
    return embed_forward_impl(e, clips, clip_index, nclips, workspace, workspace_bytes, features, nullptr, stream);
}

extern "C" int vd_embed_forward_keep(VdEmbed* e, const float* clips, const int64_t* clip_index, int64_t nclips, void* workspace,
                                     int64_t workspace_bytes, float* features, uint8_t* argmax, void* stream) {
    if (argmax == nullptr) return -1;
    return embed_forward_impl(e, clips, clip_index, nclips, workspace, workspace_bytes, features, argmax, stream);
}

// d <g_features, embed(clips)> / d clips for a vd_embed_forward_keep call (same weights): per level, last to first,
// un-pool + ReLU backward into dense dy slots (power-of-two scaled for the fp16 formats), then the level's dgrad programs
extern "C" int vd_embed_backward(VdEmbed* e, const float* g_features, const uint8_t* argmax, int64_t nclips, void* workspace,
                                 int64_t workspace_bytes, float* g_clips, void* stream) {
    if (e == nullptr || g_features == nullptr || argmax == nullptr || g_clips == nullptr || nclips < 0 || nclips > 0x7fffffff) return -1;
    if (e->prec_bwd < 0) return -8;
    if (!e->has_weights) return -6;
    if (nclips == 0) return 0;
    if (workspace == nullptr || workspace_bytes < vd_embed_backward_workspace_bytes(e, nclips)) return -7;
    const int64_t n1 = nclips * e->slots1_per_clip, n2 = nclips * e->slots2_per_clip;
    const uint8_t* am[3];
    am[0] = reinterpret_cast<const uint8_t*>(align256(const_cast<char*>(reinterpret_cast<const char*>(argmax))));
    am[1] = reinterpret_cast<const uint8_t*>(align256(const_cast<char*>(reinterpret_cast<const char*>(am[0])) + n1 * 8));
    am[2] = reinterpret_cast<const uint8_t*>(align256(const_cast<char*>(reinterpret_cast<const char*>(am[1])) + n2 * 8));
    int64_t dymax = 0;
    for (int l = 0; l < 3; ++l) dymax = dymax > dy_slots(e, l, nclips) ? dymax : dy_slots(e, l, nclips);
    char* dy = align256(static_cast<char*>(workspace));
    char* dxbuf[3] = {nullptr, nullptr, nullptr};
    char* cur = align256(dy + (int64_t)e->planes_bwd * dymax * 16);
    for (int l = 1; l < 3; ++l) {
        dxbuf[l] = cur;
        cur = align256(cur + nclips * (int64_t)e->dims[l][2] * e->dims[l][3] * e->dims[l][4] * e->dims[l][0] * 4);
    }
    float* scale = reinterpret_cast<float*>(cur);
    const bool scaled = (e->prec_bwd == VD_PREC_F16 || e->prec_bwd == VD_PREC_F16X3);
    const float* grad = g_features;
    int64_t grad_n = nclips * e->nfeat;
    int layout = 0;
    for (int l = 2; l >= 0; --l) {
        const int* d = e->dims[l];
        const int64_t nslots = dy_slots(e, l, nclips);
        float* sc = nullptr;
        int rc = 0;
        if (scaled) {
            sc = scale + 4 * l;
            rc = vd_absmax_scale(grad, grad_n, 1024.0f, sc, stream);
            if (rc) return rc;
        }
        rc = vd_unpool_relu_bwd(grad, am[l], nclips, d[1], d[8], d[9], d[10], d[11], d[5], d[6], d[7], layout, dy,
                                e->planes_bwd == 2 ? dy + nslots * 16 : nullptr, e->prec_bwd, sc, stream);
        if (rc) return rc;
        float* outp = (l == 0) ? g_clips : reinterpret_cast<float*>(dxbuf[l]);
        for (int c = 0; c < e->nbwd[l]; ++c) {
            rc = vd_program_run_scaled(e->bwd[l][c], dy, nslots, nullptr, outp, 0, nullptr, nullptr, (int)nclips, sc ? sc + 1 : nullptr, stream);
            if (rc) return rc;
        }
        grad = outp;
        grad_n = nclips * (int64_t)d[2] * d[3] * d[4] * d[0];
        layout = 1;
    }
    return 0;
}

extern "C" void vd_embed_free(VdEmbed* e) {
    if (e == nullptr) return;
    for (int l = 0; l < 3; ++l) {
        vd_program_free(e->prog[l]);
        for (int c = 0; c < 4; ++c) vd_program_free(e->bwd[l][c]);
    }
    free(e);
}

// ---- One evaluate_synset training step (utils.py:765-792, 852-853) as ONE handle ------------------------------------------
// forward with kept activations -> head -> CrossEntropy -> head backward -> per level: bias gradient, weight gradient (the
// program of vd_program_build_wgrad), input gradient -> torch.optim.SGD(momentum, weight_decay) on the 8 tensors.  The same
// call sequence as train.TrainEngine.loss_and_grads + sgd_step, from C: no Python, no torch.
struct VdTrain {
    VdEmbed* e;
    VdProgram* wg[3];
    int block[3][3], replicas[3];
    int K, kt, kh, kw, Tp;
    int64_t nclips;
    int ordered;          // vd_get_deterministic() when the handle was created: fixed-order accumulation everywhere
};

static int64_t a256(int64_t n) { return (n + 255) & ~(int64_t)255; }

extern "C" void vd_train_free(VdTrain* t) {
    if (t == nullptr) return;
    vd_embed_free(t->e);
    for (int l = 0; l < 3; ++l) vd_program_free(t->wg[l]);
    free(t);
}

extern "C" int vd_train_create(int frames, int height, int width, int num_classes, int prec, int prec_bwd, int64_t nclips, VdTrain** out) {
    if (out == nullptr || num_classes < 1 || nclips < 1 || nclips > 0x7fffffff || prec < 0 || prec > 3 || prec_bwd < 0 || prec_bwd > 3) return -1;
    const int planes = prec >= 2 ? 2 : 1, planes_bwd = prec_bwd >= 2 ? 2 : 1;
    // the gradient passes read the forward's 16-bit activations: same 16-bit family, no more planes than the forward kept
    if (planes_bwd > planes || ((prec & 1) != (prec_bwd & 1))) return -2;
    VdTrain* t = static_cast<VdTrain*>(calloc(1, sizeof(VdTrain)));
    if (t == nullptr) return -5;
    int hint = 1;
    while (hint < nclips && hint < 512) hint *= 2;
    int rc = vd_embed_create_ex(frames, height, width, prec, prec_bwd, nclips > 512 ? 0 : hint, &t->e);
    t->ordered = vd_get_deterministic();      // (vd_program_build_wgrad below reads the same switch: one copy per box)
    for (int l = 0; rc == 0 && l < 3; ++l) {
        void* blob = nullptr;
        int64_t n = 0;
        rc = vd_program_build_wgrad(l, frames, height, width, (int)nclips, planes_bwd, &blob, &n, t->block[l], &t->replicas[l]);
        if (rc == 0) { rc = vd_program_load(blob, n, prec_bwd, &t->wg[l]); vd_blob_free(blob); }
    }
    if (rc != 0) { vd_train_free(t); return rc; }
    t->K = num_classes; t->nclips = nclips;
    t->kt = 2; t->kh = height > 64 ? 2 : 1; t->kw = height > 64 ? 2 : 1;            // networks.py:733
    t->Tp = t->e->dims[2][8] - t->kt + 1;
    *out = t;
    return 0;
}

struct TrainLayout {      // byte offsets into the caller's workspace
    int64_t fwd, argmax, bwd, xT, bp, copies, feats, g_feat, dropped, logits, amt, loss, dlog, grads, scale, bias_part, total;
    int64_t gofs[8], gsz[8];
};

static TrainLayout train_layout(const VdTrain* t) {
    const VdEmbed* e = t->e;
    const int64_t B = t->nclips;
    TrainLayout L;
    int64_t o = 0;        // offsets are relative to the workspace pointer rounded UP to 256 bytes: the slack is added to `total` below
    auto take = [&](int64_t bytes) { const int64_t at = o; o += a256(bytes); return at; };
    L.fwd = take(vd_embed_workspace_bytes(e, B));
    L.argmax = take(vd_embed_argmax_bytes(e, B));
    L.bwd = take(vd_embed_backward_workspace_bytes(e, B));
    int64_t xT = 0, bp = 0, cp = 0;
    for (int l = 0; l < 3; ++l) {
        const int* d = e->dims[l];
        const int64_t CCb = (B + 7) / 8;
        xT = std::max<int64_t>(xT, (int64_t)e->planes_bwd * d[0] * CCb * d[2] * d[3] * d[4] * 16);
        const int nt = t->block[l][0], noh = t->block[l][1], now = t->block[l][2];
        const int64_t nbox = (int64_t)((d[5] + nt - 1) / nt) * ((d[6] + noh - 1) / noh) * ((d[7] + now - 1) / now);
        bp = std::max<int64_t>(bp, (int64_t)e->planes_bwd * nbox * CCb * (nt * noh * now / 2) * (d[1] / 32) * 64 * 16);
        cp = std::max<int64_t>(cp, (int64_t)t->replicas[l] * d[0] * 147 * d[1] * 4);
    }
    L.xT = take(xT); L.bp = take(bp); L.copies = take(cp);
    L.feats = take(B * e->nfeat * 4); L.g_feat = take(B * e->nfeat * 4);
    L.dropped = take(B * (int64_t)t->Tp * 128 * 4);
    L.logits = take(B * (int64_t)t->K * 4); L.amt = take(B * (int64_t)t->K * 4);
    L.loss = take(B * 4); L.dlog = take(B * (int64_t)t->K * 4);
    const int64_t sizes[8] = {64 * 3 * 147, 64, 128 * 64 * 147, 128, 128 * 128 * 147, 128, (int64_t)t->K * 128, t->K};
    L.grads = o;
    for (int i = 0; i < 8; ++i) { L.gsz[i] = sizes[i]; L.gofs[i] = take(sizes[i] * 4); }
    L.scale = take(16 * 4);
    int64_t part = 0;       // ordered mode: the bias gradients' per-workgroup partial sums
    for (int l = 0; t->ordered && l < 3; ++l) {
        const int* d = e->dims[l];
        part = std::max<int64_t>(part, vd_bias_grad_pooled_scratch_floats(B, d[1], (int64_t)d[8] * d[9] * d[10]) * 4);
    }
    L.bias_part = take(part);
    L.total = o + 256;    // room for rounding an arbitrary caller pointer up to the next 256-byte boundary
    return L;
}

extern "C" int64_t vd_train_workspace_bytes(const VdTrain* t) { return t ? train_layout(t).total : -1; }

// params[8] / momentum[8]: device fp32 tensors in parameters() order (features.0.weight, .bias, features.3.*, features.6.*,
// logit.weight (K,128,1,1,1), logit.bias), updated IN PLACE; `first` != 0: the momentum buffers are initialised with the
// gradient (torch.optim.SGD's first step).  clips (nclips, T, 3, H, W) fp32 -- already standardised (vd_standardize) as
// epoch() does; labels int64; dropout_mask (nclips, 128, T') holding 0 or 1/(1-p), or NULL (eval-style, p = 0).
// Outputs (device, optional): loss_per_clip (nclips), logits (nclips, K).
extern "C" int vd_train_step(VdTrain* t, float* const* params, float* const* momentum, const float* clips, const int64_t* labels,
                             const float* dropout_mask, float lr, float mom, float weight_decay, int first, void* workspace,
                             int64_t workspace_bytes, float* loss_per_clip, float* logits_out, void* stream) {
    if (t == nullptr || params == nullptr || momentum == nullptr || clips == nullptr || labels == nullptr) return -1;
    for (int i = 0; i < 8; ++i) if (params[i] == nullptr || momentum[i] == nullptr) return -1;
    const TrainLayout L = train_layout(t);
    if (workspace == nullptr || workspace_bytes < L.total) return -7;
    VdEmbed* e = t->e;
    const int64_t B = t->nclips;
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* feats = reinterpret_cast<float*>(ws + L.feats);
    float* g_feat = reinterpret_cast<float*>(ws + L.g_feat);
    float* dropped = reinterpret_cast<float*>(ws + L.dropped);
    float* logits = reinterpret_cast<float*>(ws + L.logits);
    int32_t* amt = reinterpret_cast<int32_t*>(ws + L.amt);
    float* loss_c = reinterpret_cast<float*>(ws + L.loss);
    float* dlog = reinterpret_cast<float*>(ws + L.dlog);
    float* g[8];
    for (int i = 0; i < 8; ++i) g[i] = reinterpret_cast<float*>(ws + L.gofs[i]);
    uint8_t* argmax = reinterpret_cast<uint8_t*>(ws + L.argmax);
    int rc = vd_embed_set_weights(e, params[0], params[1], params[2], params[3], params[4], params[5], stream);
    if (rc == 0) rc = vd_embed_forward_keep(e, clips, nullptr, B, ws + L.fwd, vd_embed_workspace_bytes(e, B), feats, argmax, stream);
    const int* d2 = e->dims[2];
    if (rc == 0) rc = vd_head_train_fwd(feats, dropout_mask, params[6], params[7], B, d2[1], d2[8], d2[9], d2[10], t->kt, t->kh, t->kw, t->K,
                                        dropped, logits, amt, stream);
    if (rc == 0) rc = vd_ce_loss(logits, labels, (int)B, t->K, loss_c, dlog, stream);
    if (rc) return rc;
    if (hipMemsetAsync(ws + L.grads, 0, (size_t)(L.scale - L.grads), st) != hipSuccess) return -9;
    rc = (t->ordered ? vd_head_train_bwd_ordered : vd_head_train_bwd)(dlog, amt, dropped, dropout_mask, params[6], B, d2[1], d2[8], d2[9],
                                                                      d2[10], t->kt, t->kh, t->kw, t->K, g[6], g[7], g_feat, stream);
    if (rc) return rc;
    // kept activations and arg-max bytes: the layout of embed_forward_impl
    const int64_t n0 = B * e->slots0_per_clip, n1 = B * e->slots1_per_clip, n2 = B * e->slots2_per_clip;
    char* rows = align256(ws + L.fwd);
    char* act[3] = {nullptr, align256(rows + (int64_t)e->planes * n0 * 16), nullptr};
    act[2] = align256(act[1] + (int64_t)e->planes * n1 * 16);
    const int64_t act_plane[3] = {0, n1, n2};
    const uint8_t* am[3];
    am[0] = reinterpret_cast<const uint8_t*>(align256(reinterpret_cast<char*>(argmax)));
    am[1] = reinterpret_cast<const uint8_t*>(align256(const_cast<char*>(reinterpret_cast<const char*>(am[0])) + n1 * 8));
    am[2] = reinterpret_cast<const uint8_t*>(align256(const_cast<char*>(reinterpret_cast<const char*>(am[1])) + n2 * 8));
    // dense dy + dx buffers: the layout of vd_embed_backward
    int64_t dymax = 0;
    for (int l = 0; l < 3; ++l) dymax = std::max(dymax, dy_slots(e, l, B));
    char* dy = align256(ws + L.bwd);
    char* dxbuf[3] = {nullptr, nullptr, nullptr};
    char* cur = align256(dy + (int64_t)e->planes_bwd * dymax * 16);
    for (int l = 1; l < 3; ++l) {
        dxbuf[l] = cur;
        cur = align256(cur + B * (int64_t)e->dims[l][2] * e->dims[l][3] * e->dims[l][4] * e->dims[l][0] * 4);
    }
    float* scale = reinterpret_cast<float*>(ws + L.scale);
    const bool scaled = (e->prec_bwd == VD_PREC_F16 || e->prec_bwd == VD_PREC_F16X3);
    char* xT = ws + L.xT;
    char* bp = ws + L.bp;
    float* copies = reinterpret_cast<float*>(ws + L.copies);
    const float* grad = g_feat;
    int64_t grad_n = B * e->nfeat;
    int layout = 0;
    for (int l = 2; l >= 0; --l) {
        const int* d = e->dims[l];
        const int cin = d[0], cout = d[1];
        const int64_t nslots = dy_slots(e, l, B), CCb = (B + 7) / 8, npos_in = (int64_t)d[2] * d[3] * d[4];
        float* sc = nullptr;
        if (scaled) {
            sc = scale + 4 * l;
            if ((rc = vd_absmax_scale(grad, grad_n, 1024.0f, sc, stream))) return rc;
        }
        if (l > 0) {
            rc = vd_unpool_relu_bwd(grad, am[l], B, cout, d[8], d[9], d[10], d[11], d[5], d[6], d[7], layout, dy,
                                    e->planes_bwd == 2 ? dy + nslots * 16 : nullptr, e->prec_bwd, sc, stream);
            if (rc) return rc;
        }
        if (t->ordered) rc = vd_bias_grad_pooled_ordered(grad, am[l], B, cout, (int64_t)d[8] * d[9] * d[10], layout,
                                                         reinterpret_cast<float*>(ws + L.bias_part), g[2 * l + 1], stream);
        else rc = vd_bias_grad_pooled(grad, am[l], B, cout, (int64_t)d[8] * d[9] * d[10], layout, g[2 * l + 1], stream);
        if (rc) return rc;
        // weight gradient: x clip-minor, dy packed straight from the pooled gradient, boxes accumulate into copies
        const int64_t xT_plane = (int64_t)cin * CCb * npos_in;
        if (l == 0) rc = vd_clip_minor_pix(clips, B, d[2], d[3], d[4], xT, e->planes_bwd == 2 ? xT + xT_plane * 16 : nullptr, e->prec_bwd, stream);
        else rc = vd_clip_minor_cl(act[l], act_plane[l], e->planes_bwd, B, cin, npos_in, xT, xT_plane, stream);
        if (rc) return rc;
        const int nt = t->block[l][0], noh = t->block[l][1], now = t->block[l][2];
        const int64_t nbox = (int64_t)((d[5] + nt - 1) / nt) * ((d[6] + noh - 1) / noh) * ((d[7] + now - 1) / now);
        const int64_t bp_elems = nbox * CCb * (nt * noh * now / 2) * (cout / 32) * 64 * 8;
        rc = vd_unpool_relu_bwd_packed(grad, am[l], B, cout, d[8], d[9], d[10], d[11], d[5], d[6], d[7], layout, nt, noh, now, bp,
                                       e->planes_bwd == 2 ? bp + bp_elems * 2 : nullptr, e->prec_bwd, sc, stream);
        if (rc) return rc;
        const int64_t copy_elems = (int64_t)cin * 147 * cout;
        if (hipMemsetAsync(copies, 0, (size_t)t->replicas[l] * copy_elems * 4, st) != hipSuccess) return -9;
        rc = vd_program_run_wgrad(t->wg[l], xT, xT_plane, bp, bp_elems, copies, copy_elems, cin, sc ? sc + 1 : nullptr, stream);
        if (rc == 0) rc = vd_replica_sum(copies, t->replicas[l], cin * 147, cout, g[2 * l], stream);
        if (rc) return rc;
        if (l > 0) {
            float* outp = reinterpret_cast<float*>(dxbuf[l]);
            for (int c = 0; c < e->nbwd[l]; ++c)
                if ((rc = vd_program_run_scaled(e->bwd[l][c], dy, nslots, nullptr, outp, 0, nullptr, nullptr, (int)B, sc ? sc + 1 : nullptr, stream))) return rc;
            grad = outp;
            grad_n = B * npos_in * cin;
            layout = 1;
        }
    }
    for (int i = 0; i < 8; ++i)
        if ((rc = vd_sgd_momentum_wd(params[i], momentum[i], g[i], L.gsz[i], lr, mom, weight_decay, first, stream))) return rc;
    if (loss_per_clip != nullptr && hipMemcpyAsync(loss_per_clip, loss_c, (size_t)B * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return -9;
    if (logits_out != nullptr && hipMemcpyAsync(logits_out, logits, (size_t)B * t->K * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return -9;
    return 0;
}
