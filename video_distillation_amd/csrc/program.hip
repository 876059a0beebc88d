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
       H_SRC_CHUNK4, H_MT_VALID, H_N_DESC, H_N_TABLES, H_N_BOXES, H_N_GATHER, H_N_WIDX, H_N_COLOFF, H_PERSIST, H_WORDS = 40 };

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
        if (g->d_tables) hipFree(g->d_tables);
        if (g->d_zero) hipFree(g->d_zero);
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
    if (g == nullptr || src == nullptr || dst == nullptr || nclips < 0) return -1;
    VdConvParams p = g->p;         // per-call copy: the handle may be used from several streams
    p.src = src; p.src_plane_stride4 = src_plane_slots * 4;
    p.bias = bias; p.dst = dst; p.dst_plane_stride = dst_plane_stride;
    p.argmax = argmax; p.clip_index = clip_index; p.nclips = nclips;
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
    if (g->d_tables) hipFree(g->d_tables);
    if (g->d_zero) hipFree(g->d_zero);
    if (g->d_wpk) hipFree(g->d_wpk);
    free(g);
}
