// Stand-ins for the kernel entry points the handle code of csrc/program.hip calls (defined for real in csrc/conv_mfma.hip and
// csrc/aux_kernels.hip): each one READS the whole extent of its inputs and WRITES the whole extent of its outputs, as given by
// its documented argument contract (include/vd_hip.h) -- so that under AddressSanitizer any buffer the workspace-layout code
// made too small, misplaced or left misaligned shows up as an out-of-bounds access, exactly where the real kernel would have
// corrupted device memory silently.  No arithmetic.  Test infrastructure only (tests/native/Makefile).
#include <stdint.h>
#include <string.h>

#include "../../include/vd_hip.h"

namespace {
volatile uint64_t g_sink;
int64_t g_bytes_read, g_bytes_written;

void rd(const void* p, int64_t bytes) {
    if (p == nullptr || bytes <= 0) return;
    const unsigned char* c = static_cast<const unsigned char*>(p);
    uint64_t s = 0;
    for (int64_t i = 0; i < bytes; i += 64) s += c[i];
    s += c[bytes - 1];
    g_sink += s;
    g_bytes_read += bytes;
}
void wr(void* p, int64_t bytes, int v = 0x5a) {
    if (p == nullptr || bytes <= 0) return;
    memset(p, v, (size_t)bytes);
    g_bytes_written += bytes;
}
bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
int planes_of(int prec) { return prec >= 2 ? 2 : 1; }
}  // namespace

extern "C" int64_t vd_stub_bytes(int what) { return what ? g_bytes_written : g_bytes_read; }

extern "C" int vd_conv_mfma(const VdConvParams* pp, void*) {
    if (pp == nullptr) return -1;
    const VdConvParams& p = *pp;
    const int planes = planes_of(p.prec);
    if (!aligned(p.src, 4) || !aligned(p.wpk, 16) || !aligned(p.dst, 4)) return -90;        // what the kernel's vector accesses assume
    for (int pl = 0; pl < planes; ++pl) {
        rd(static_cast<const char*>(p.src) + (int64_t)pl * p.src_plane_stride4 * 4, (int64_t)p.nclips * p.src_clip_stride4 * 4);
        const int64_t welems = p.w_box_stride ? (int64_t)p.nbox * p.w_box_stride : p.w_plane_stride;
        rd(static_cast<const char*>(p.wpk) + (int64_t)pl * p.w_plane_stride * 2, welems * 2);
    }
    rd(p.boxes, (int64_t)p.nbox * 8 * 4);
    rd(p.gather, (int64_t)p.nbox * p.gather_stride * 4);
    rd(p.zero_slot, 16);
    if (p.bias) rd(p.bias, (int64_t)p.n_out * 4);
    if (p.out_scale) rd(p.out_scale, 4);
    if (p.epi == VD_EPI_POOL_CL) {
        if (!aligned(p.dst, 16)) return -90;
        const int oplanes = (planes == 2 || p.emit_lo) ? 2 : 1;
        for (int pl = 0; pl < oplanes; ++pl)
            wr(static_cast<char*>(p.dst) + (int64_t)pl * p.dst_plane_stride * 16, (int64_t)p.nclips * p.out_clip_stride * 16);
        if (p.argmax) (p.select ? rd(p.argmax, (int64_t)p.nclips * p.out_clip_stride * 8) : wr(p.argmax, (int64_t)p.nclips * p.out_clip_stride * 8));
    } else if (p.epi == VD_EPI_POOL_FEAT) {
        wr(p.dst, (int64_t)p.nclips * p.out_clip_stride * 4);
        if (p.argmax) wr(p.argmax, (int64_t)p.nclips * p.out_clip_stride);
    } else {
        if (p.atomic && p.replica_stride > 0) {
            int rmax = 0;
            for (int b = 0; b < p.nbox; ++b) rmax = p.boxes[b * 8 + 5] > rmax ? p.boxes[b * 8 + 5] : rmax;
            // (accumulation: read-modify-write of every copy)
            rd(p.dst, (int64_t)(rmax + 1) * p.replica_stride * 4);
            wr(p.dst, (int64_t)(rmax + 1) * p.replica_stride * 4, 0);
        } else {
            wr(p.dst, (int64_t)p.nclips * p.out_clip_stride * 4);
        }
    }
    return 0;
}
extern "C" int vd_conv0_breg(const VdConvParams* p, void* s) { return vd_conv_mfma(p, s); }

extern "C" int vd_pack_weights(const float* w, const int32_t* widx, int64_t n, void* hi, void* lo, int, void*) {
    rd(widx, n * 4);
    int64_t mx = -1;
    for (int64_t i = 0; i < n; ++i) mx = widx[i] > mx ? widx[i] : mx;
    rd(w, (mx + 1) * 4);
    wr(hi, n * 2); wr(lo, n * 2);
    return 0;
}

extern "C" int vd_pix2rows(const float* x, const int64_t* clip_index, int64_t nclips, int T, int H, int W, void* hi, void* lo, int, void*) {
    const int64_t pitch = ((W + 8 + 7) / 8) * 8;
    if (!aligned(hi, 16)) return -90;
    if (clip_index) rd(clip_index, nclips * 8);
    else rd(x, nclips * T * 3 * (int64_t)H * W * 4);
    wr(hi, nclips * T * 3 * (int64_t)H * pitch * 2); wr(lo, nclips * T * 3 * (int64_t)H * pitch * 2);
    return 0;
}

extern "C" int vd_unpool_relu_bwd(const float* g, const uint8_t* argmax, int64_t nclips, int C, int To, int Ho, int Wo, int, int T,
                                  int OH, int OW, int, void* hi, void* lo, int, const float* scale, void*) {
    if (!aligned(hi, 16) || !aligned(g, 4)) return -90;
    rd(g, nclips * C * (int64_t)To * Ho * Wo * 4);
    rd(argmax, nclips * C * (int64_t)To * Ho * Wo);
    if (scale) rd(scale, 4);
    const int64_t bytes = nclips * (C / 8) * (int64_t)T * OH * OW * 16;
    wr(hi, bytes); wr(lo, bytes);
    return 0;
}

extern "C" int vd_unpool_relu_bwd_packed(const float* g, const uint8_t* argmax, int64_t nclips, int C, int To, int Ho, int Wo, int,
                                         int T, int OH, int OW, int, int nt, int noh, int now, void* hi, void* lo, int,
                                         const float* scale, void*) {
    if (!aligned(hi, 16)) return -90;
    rd(g, nclips * C * (int64_t)To * Ho * Wo * 4);
    rd(argmax, nclips * C * (int64_t)To * Ho * Wo);
    if (scale) rd(scale, 4);
    const int64_t nbox = (int64_t)((T + nt - 1) / nt) * ((OH + noh - 1) / noh) * ((OW + now - 1) / now);
    const int64_t elems = nbox * ((nclips + 7) / 8) * (nt * noh * now / 2) * (C / 32) * 64 * 8;
    wr(hi, elems * 2); wr(lo, elems * 2);
    return 0;
}

extern "C" int vd_absmax_scale(const float* x, int64_t n, float, float* out4, void*) {
    if (!aligned(out4, 4)) return -90;
    rd(x, n * 4); wr(out4, 16);
    return 0;
}

extern "C" int vd_head_train_fwd(const float* feats, const float* mask, const float* w, const float* b, int64_t nclips, int C, int To,
                                 int Ho, int Wo, int kt, int, int, int K, float* dropped, float* logits, int32_t* amax_t, void*) {
    const int Tp = To - kt + 1;
    rd(feats, nclips * C * (int64_t)To * Ho * Wo * 4);
    if (mask) rd(mask, nclips * C * (int64_t)Tp * 4);
    rd(w, (int64_t)K * C * 4); rd(b, (int64_t)K * 4);
    wr(dropped, nclips * Tp * (int64_t)C * 4); wr(logits, nclips * K * 4); wr(amax_t, nclips * K * 4, 0);
    return 0;
}

extern "C" int vd_ce_loss(const float* logits, const int64_t* labels, int B, int K, float* loss, float* dlog, void*) {
    if (!aligned(labels, 8)) return -90;
    rd(logits, (int64_t)B * K * 4); rd(labels, (int64_t)B * 8);
    wr(loss, (int64_t)B * 4); wr(dlog, (int64_t)B * K * 4);
    return 0;
}

static int head_bwd(const float* dlog, const int32_t* amt, const float* dropped, const float* mask, const float* w, int64_t nclips, int C,
                    int To, int Ho, int Wo, int kt, int K, float* g_w, float* g_b, float* g_feats) {
    const int Tp = To - kt + 1;
    rd(dlog, nclips * K * 4); rd(amt, nclips * K * 4); rd(dropped, nclips * Tp * (int64_t)C * 4);
    if (mask) rd(mask, nclips * C * (int64_t)Tp * 4);
    rd(w, (int64_t)K * C * 4);
    rd(g_w, (int64_t)K * C * 4); wr(g_w, (int64_t)K * C * 4); rd(g_b, (int64_t)K * 4); wr(g_b, (int64_t)K * 4);
    wr(g_feats, nclips * C * (int64_t)To * Ho * Wo * 4);
    return 0;
}
extern "C" int vd_head_train_bwd(const float* a, const int32_t* b, const float* c, const float* d, const float* e, int64_t n, int C, int To,
                                 int Ho, int Wo, int kt, int, int, int K, float* g_w, float* g_b, float* g_feats, void*) {
    return head_bwd(a, b, c, d, e, n, C, To, Ho, Wo, kt, K, g_w, g_b, g_feats);
}
extern "C" int vd_head_train_bwd_ordered(const float* a, const int32_t* b, const float* c, const float* d, const float* e, int64_t n, int C,
                                         int To, int Ho, int Wo, int kt, int, int, int K, float* g_w, float* g_b, float* g_feats, void*) {
    return head_bwd(a, b, c, d, e, n, C, To, Ho, Wo, kt, K, g_w, g_b, g_feats);
}

extern "C" int vd_bias_grad_pooled(const float* g, const uint8_t* argmax, int64_t nclips, int C, int64_t npos, int, float* db, void*) {
    rd(g, nclips * C * npos * 4); rd(argmax, nclips * C * npos); rd(db, (int64_t)C * 4); wr(db, (int64_t)C * 4);
    return 0;
}
extern "C" int64_t vd_bias_grad_pooled_scratch_floats(int64_t nclips, int C, int64_t npos) {
    if (nclips <= 0 || npos <= 0 || C <= 0) return 0;
    return nclips * ((npos + 255) / 256) * C;
}
extern "C" int vd_bias_grad_pooled_ordered(const float* g, const uint8_t* argmax, int64_t nclips, int C, int64_t npos, int, float* scratch,
                                           float* db, void*) {
    if (!aligned(scratch, 4)) return -90;
    rd(g, nclips * C * npos * 4); rd(argmax, nclips * C * npos);
    wr(scratch, vd_bias_grad_pooled_scratch_floats(nclips, C, npos) * 4);
    rd(db, (int64_t)C * 4); wr(db, (int64_t)C * 4);
    return 0;
}

extern "C" int vd_clip_minor_pix(const float* x, int64_t nclips, int T, int H, int W, void* hi, void* lo, int, void*) {
    if (!aligned(hi, 16)) return -90;
    rd(x, nclips * T * 3 * (int64_t)H * W * 4);
    const int64_t bytes = 3 * ((nclips + 7) / 8) * (int64_t)T * H * W * 16;
    wr(hi, bytes); wr(lo, bytes);
    return 0;
}
extern "C" int vd_clip_minor_cl(const void* src, int64_t src_plane_slots, int planes, int64_t nclips, int C, int64_t npos, void* dst,
                                int64_t dst_plane_slots, void*) {
    if (!aligned(dst, 16) || !aligned(src, 16)) return -90;
    for (int pl = 0; pl < planes; ++pl) {
        rd(static_cast<const char*>(src) + pl * src_plane_slots * 16, nclips * (C / 8) * npos * 16);
        wr(static_cast<char*>(dst) + pl * dst_plane_slots * 16, (int64_t)C * ((nclips + 7) / 8) * npos * 16);
    }
    return 0;
}

extern "C" int vd_replica_sum(float* rep, int replicas, int rows, int cols, float* out, void*) {
    rd(rep, (int64_t)replicas * rows * cols * 4); wr(rep, (int64_t)replicas * rows * cols * 4, 0);       // (the copies are scratch: folded in place)
    rd(out, (int64_t)rows * cols * 4); wr(out, (int64_t)rows * cols * 4);
    return 0;
}

extern "C" int vd_sgd_momentum_wd(float* x, float* buf, const float* g, int64_t n, float, float, float, int first, void*) {
    rd(x, n * 4); rd(g, n * 4);
    if (!first) rd(buf, n * 4);
    wr(buf, n * 4); wr(x, n * 4);
    return 0;
}
