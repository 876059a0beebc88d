// Host-side planner of the forward and input-gradient tile programs, in C++: the C ABI needs no Python step.
//
// What video_distillation_amd/plan.py does for the three forward programs of one ConvNet3D geometry
// (plan_forward_pix, plan_forward_cl, the wave-layout choice of plan_network and latency_variant), re-stated in C++ with
// the same search order and tie-breaking, so that vd_program_build() emits byte for byte the blob plan.export_program()
// writes (tests/test_cplanner.py compares them for several geometries).  The device kernel knows nothing about
// convolution geometry; everything index-heavy is decided here:
//   * the box of conv rows a workgroup owns (fewest boxes under the row / LDS / LDS-DMA budgets),
//   * the LDS pitches of its input patch and which pool windows share an MFMA tile, searched for conflict-free
//     ds_read_b128 (MI355X LDS lane groups),
//   * gather table of the patch (LDS-DMA source offsets), tap offsets, output offsets, weight gather index.
// Geometry: Conv3d k(3,7,7) s(1,2,2) p(1,3,3) -> ReLU -> MaxPool3d (reference networks.py:792-814).
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <set>
#include <string>
#include <vector>

#include "../../include/vd_hip.h"

namespace {

constexpr int SLOT_BYTES = 16;
constexpr int EPI_POOL_CL = 0, EPI_POOL_FEAT = 1, EPI_ROWS = 2;
constexpr int KT = 3, KH = 7, KW = 7;
// ds_read_b128 services a wave in four 16-lane groups (two per lane half)
const int B128_GROUPS[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};

struct BoxType {
    int pf = 0, ph = 0, pw = 0, pitch_h = 0, pitch_f = 0, pitch_c = 0, mt = 0;
    std::vector<int32_t> a_off, out, tap_off;
    double cyc = 0.0;
};

struct Plan {
    int CC = 0, F = 0, H = 0, W = 0, NT = 0, MW = 0, MTW = 0, S = 0, ncl = 1;
    std::vector<std::array<int64_t, 6>> boxes;
    std::vector<BoxType> types;
    std::vector<int32_t> widx;
    int epi = 0, pool_t = 0, relu = 1, n_out = 0, n_stride = 0;
    int64_t out_clip_stride = 0;
    int out_chunk_stride = 0, out_t_stride = 0;
    int pair_flip = 0;                 // plan.ConvPlan.pair_flip (frame-tile programs)
    int w_step4 = 4, row_pitch4 = 0;
    int64_t clip_stride4 = 0, chunk_stride4 = 0;
    int NTW = 1;
    int atomic = 0;                    // EPI_ROWS: accumulate with fp32 atomics (weight-gradient programs)
    int64_t w_box_stride = 0;          // per-box B operand (packed dy), elements
    int64_t rows_total = 0;
    std::vector<int32_t> col_off;      // EPI_ROWS: element offset of output column n (empty: n * n_stride)
    int nbox() const { return (int)boxes.size(); }
    int64_t lds_slots() const {
        int64_t m = 0;
        for (const auto& t : types) m = std::max<int64_t>(m, (int64_t)t.pitch_c * ncl);
        return m;
    }
    int64_t grid(int64_t nclips) const { return ((nclips + ncl - 1) / ncl) * nbox(); }
};

using Tap = std::array<int, 3>;
using OutFn = std::function<int64_t(int, int, int, int)>;

double conflict_cycles(const int64_t* slots32) {
    int total = 0;
    for (int g = 0; g < 2; ++g) {
        int cnt[16] = {0};
        int mx = 0;
        for (int k = 0; k < 16; ++k) {
            const int b = (int)(((slots32[B128_GROUPS[g][k]] % 16) + 16) % 16);
            mx = std::max(mx, ++cnt[b]);
        }
        total += mx;
    }
    return 2.0 * total;
}

inline int row_of(int q, int j) { return (j & 3) + 4 * (q & 1) + 8 * (j >> 2) + 16 * (q >> 1); }
inline double round3(double x) { return std::nearbyint(x * 1000.0) / 1000.0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int FRAME_TILE_FLIP = 6, FRAME_TILE_OUT_STEP = 2;      // plan.FRAME_TILE_FLIP / FRAME_TILE_OUT_STEP
const int ORDERS[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};

std::vector<std::array<int, 4>> perms4() {
    std::vector<std::array<int, 4>> out;
    std::array<int, 4> p = {0, 1, 2, 3};
    do out.push_back(p); while (std::next_permutation(p.begin(), p.end()));   // lexicographic = itertools.permutations
    return out;
}

// One BoxType for a fixed enumeration order of the pool windows: searches the LDS pitches and, per MFMA tile, which
// window sits in which lane-half / register-half, for the fewest ds_read_b128 bank conflicts.
// plain-row programs (input-gradient passes): rows in (clip, a, b, c) order, 32 per MFMA tile, no tile assignment to search
BoxType build_type_rows(const int box[3], const int row_stride[3], const std::vector<Tap>& taps, const int ext[3], int mt_pad,
                        const OutFn& out_fn, const int valid[3], int ncl, int64_t slot_cap) {
    const int na = box[0], nb = box[1], nc = box[2];
    const int pf = ext[0], ph = ext[1], pw = ext[2];
    std::vector<std::array<int64_t, 4>> rcoord;
    std::vector<int64_t> rout;
    for (int ci = 0; ci < ncl; ++ci)
        for (int a = 0; a < na; ++a)
            for (int b = 0; b < nb; ++b)
                for (int c = 0; c < nc; ++c) {
                    rcoord.push_back({ci, (int64_t)row_stride[0] * a, (int64_t)row_stride[1] * b, (int64_t)row_stride[2] * c});
                    rout.push_back((a < valid[0] && b < valid[1] && c < valid[2]) ? out_fn(ci, a, b, c) : -1);
                }
    const int nrows = (int)rcoord.size();
    const int ntile_used = (int)cdiv(nrows, 32);
    bool have = false;
    double best_cyc = 0.0;
    int best_pitch_c = 0;
    BoxType best;
    for (int dph = 0; dph < 16; ++dph) {
        const int pitch_h = pw + dph;
        for (int dpf = 0; dpf < 16; ++dpf) {
            const int pitch_f = ph * pitch_h + dpf;
            const int pitch_c = pf * pitch_f;
            if ((int64_t)pitch_c * ncl > slot_cap && have) continue;
            std::vector<int64_t> a_off((size_t)mt_pad * 32, 0), out((size_t)mt_pad * 32, -1);
            for (int r = 0; r < nrows; ++r) {
                a_off[r] = rcoord[r][0] * pitch_c + rcoord[r][1] * pitch_f + rcoord[r][2] * pitch_h + rcoord[r][3];
                out[r] = rout[r];
            }
            double cyc_sum = 0.0;
            for (int t = 0; t < ntile_used; ++t) cyc_sum += conflict_cycles(&a_off[(size_t)t * 32]);
            const double cyc = cyc_sum / ntile_used;
            const double kc = round3(cyc);
            if (!have || kc < best_cyc || (kc == best_cyc && pitch_c < best_pitch_c)) {
                have = true; best_cyc = kc; best_pitch_c = pitch_c;
                best = BoxType();
                best.pf = pf; best.ph = ph; best.pw = pw; best.pitch_h = pitch_h; best.pitch_f = pitch_f; best.pitch_c = pitch_c;
                best.mt = mt_pad; best.cyc = cyc;
                best.a_off.resize(a_off.size()); best.out.resize(out.size());
                for (size_t k = 0; k < a_off.size(); ++k) best.a_off[k] = (int32_t)(a_off[k] * SLOT_BYTES);
                for (size_t k = 0; k < out.size(); ++k) best.out[k] = (int32_t)out[k];
                best.tap_off.resize(taps.size());
                for (size_t k = 0; k < taps.size(); ++k)
                    best.tap_off[k] = (int32_t)((taps[k][0] * pitch_f + taps[k][1] * pitch_h + taps[k][2]) * SLOT_BYTES);
            }
            if (cyc <= 4.0 + 1e-9) break;
        }
        if (have && best_cyc <= 4.0 + 1e-9) break;
    }
    return best;
}

BoxType build_type_order(const int box[3], const int row_stride[3], const std::vector<Tap>& taps, const int ext[3], int mt_pad,
                         const OutFn& out_fn, const int valid[3], int ncl, int64_t slot_cap, const int order[3]) {
    static const std::vector<std::array<int, 4>> PERMS = perms4();
    const int na = box[0], nb = box[1], nc = box[2];
    const int pf = ext[0], ph = ext[1], pw = ext[2];
    struct G { int ci, a, b, c; };
    std::vector<G> groups;
    const int lim[3] = {na, nb, nc};
    const bool frame = order == nullptr;      // plan._build_type_order(order=None): FRAME TILES of the first level
    if (frame) {
        // a row group = the pool windows (b, c) and (b, c + 4) of one frame, c in {0, 2}; a tile = the four groups of one
        // column class in one frame; the tiles of a wave row = consecutive frames
        for (int ci = 0; ci < ncl; ++ci)
            for (int c = 0; c < 4; c += 2)
                for (int a = 0; a < na; ++a)
                    for (int b = 0; b < nb; b += 2) groups.push_back({ci, a, b, c});
    } else {
        for (int ci = 0; ci < ncl; ++ci)
            for (int x = 0; x < lim[order[0]]; x += 2)
                for (int y = 0; y < lim[order[1]]; y += 2)
                    for (int z = 0; z < lim[order[2]]; z += 2) {
                        int abc[3];
                        abc[order[0]] = x; abc[order[1]] = y; abc[order[2]] = z;
                        groups.push_back({ci, abc[0], abc[1], abc[2]});
                    }
    }
    const int ngr = (int)groups.size();
    std::vector<std::array<std::array<int64_t, 4>, 8>> gcoord(ngr);
    std::vector<int64_t> gout(ngr);
    for (int gi = 0; gi < ngr; ++gi) {
        const G& g = groups[gi];
        for (int j = 0; j < 8; ++j) {
            int dt = (j >> 2) & 1, dh = (j >> 1) & 1, dw = j & 1;
            if (frame) { dw += 4 * dt; dt = 0; }
            gcoord[gi][j] = {g.ci, (int64_t)row_stride[0] * (g.a + dt), (int64_t)row_stride[1] * (g.b + dh),
                             (int64_t)row_stride[2] * (g.c + dw)};
        }
        const bool ok = frame ? (g.a < valid[0] && g.b + 2 <= valid[1])
                              : (g.a + 2 <= valid[0] && g.b + 2 <= valid[1] && g.c + 2 <= valid[2]);
        gout[gi] = ok ? out_fn(g.ci, g.a, g.b, g.c) : -1;
    }
    const int ntile_used = (int)cdiv(ngr, 4);
    bool have = false;
    double best_cyc = 0.0;
    int best_pitch_c = 0;
    BoxType best;
    for (int dph = 0; dph < 16; ++dph) {
        const int pitch_h = pw + dph;
        for (int dpf = 0; dpf < 16; ++dpf) {
            const int pitch_f = ph * pitch_h + dpf;
            const int pitch_c = pf * pitch_f;
            if ((int64_t)pitch_c * ncl > slot_cap && have) continue;
            std::vector<int64_t> a_off((size_t)mt_pad * 32, 0), out((size_t)mt_pad * 4, -1);
            double cyc_sum = 0.0;
            for (int tile = 0; tile < ntile_used; ++tile) {
                int gids[4];
                int cnt = 0;
                for (int g = tile * 4; g < std::min(ngr, tile * 4 + 4); ++g) gids[cnt++] = g;
                for (int k = cnt; k < 4; ++k) gids[k] = gids[0];          // pad with a duplicate (discarded)
                bool have_t = false;
                double bt_cyc = 0.0;
                std::array<int, 4> bt_perm = {0, 1, 2, 3};
                int64_t bt_slots[32];
                for (const auto& perm : PERMS) {
                    int64_t slots[32];
                    for (int q = 0; q < 4; ++q) {
                        const auto& gc = gcoord[gids[perm[q]]];
                        const int jx = (frame && ((FRAME_TILE_FLIP >> q) & 1)) ? 4 : 0;      // this group lists its second window first
                        for (int j = 0; j < 8; ++j) {
                            const auto& rc = gc[j ^ jx];
                            slots[row_of(q, j)] = rc[0] * pitch_c + rc[1] * pitch_f + rc[2] * pitch_h + rc[3];
                        }
                    }
                    const double cyc = conflict_cycles(slots);
                    if (!have_t || cyc < bt_cyc) {
                        have_t = true; bt_cyc = cyc; bt_perm = perm;
                        memcpy(bt_slots, slots, sizeof(slots));
                    }
                    if (cyc <= 4.0) break;
                }
                cyc_sum += bt_cyc;
                for (int r = 0; r < 32; ++r) a_off[(size_t)tile * 32 + r] = bt_slots[r];
                for (int q = 0; q < 4; ++q) {
                    const int gi = tile * 4 + bt_perm[q];
                    int64_t o = gi < ngr ? gout[gi] : -1;
                    if (frame && ((FRAME_TILE_FLIP >> q) & 1) && o >= 0) o += FRAME_TILE_OUT_STEP;   // first window = the right-hand one
                    out[(size_t)tile * 4 + q] = o;
                }
            }
            const double cyc = cyc_sum / ntile_used;
            const double kc = round3(cyc);
            if (!have || kc < best_cyc || (kc == best_cyc && pitch_c < best_pitch_c)) {
                have = true; best_cyc = kc; best_pitch_c = pitch_c;
                best = BoxType();
                best.pf = pf; best.ph = ph; best.pw = pw; best.pitch_h = pitch_h; best.pitch_f = pitch_f; best.pitch_c = pitch_c;
                best.mt = mt_pad; best.cyc = cyc;
                best.a_off.resize(a_off.size()); best.out.resize(out.size());
                for (size_t k = 0; k < a_off.size(); ++k) best.a_off[k] = (int32_t)(a_off[k] * SLOT_BYTES);
                for (size_t k = 0; k < out.size(); ++k) best.out[k] = (int32_t)out[k];
                best.tap_off.resize(taps.size());
                for (size_t k = 0; k < taps.size(); ++k)
                    best.tap_off[k] = (int32_t)((taps[k][0] * pitch_f + taps[k][1] * pitch_h + taps[k][2]) * SLOT_BYTES);
            }
            if (cyc <= 4.0 + 1e-9) break;
        }
        if (have && best_cyc <= 4.0 + 1e-9) break;
    }
    return best;
}

BoxType build_type(const int box[3], const int row_stride[3], const std::vector<Tap>& taps, const int ext[3], int mt_pad,
                   const OutFn& out_fn, const int valid[3], int ncl, int64_t slot_cap, bool frame_tiles = false) {
    if (frame_tiles) return build_type_order(box, row_stride, taps, ext, mt_pad, out_fn, valid, ncl, slot_cap, nullptr);
    bool have = false;
    double bk = 0.0;
    int bp = 0;
    BoxType best;
    for (const auto& order : ORDERS) {
        BoxType bt = build_type_order(box, row_stride, taps, ext, mt_pad, out_fn, valid, ncl, slot_cap, order);
        const double k = round3(bt.cyc);
        const bool stop = bt.cyc <= 4.0 + 1e-9 && (int64_t)bt.pitch_c * ncl <= slot_cap;
        if (!have || k < bk || (k == bk && bt.pitch_c < bp)) { have = true; bk = k; bp = bt.pitch_c; best = std::move(bt); }
        if (stop) break;
    }
    return best;
}

struct PlanSpec {
    int src_grid[3];
    int CC;
    int row_dims[3];
    int row_origin[3];
    int row_stride[3];
    std::vector<Tap> taps;
    int n_out, NT, MW;
    std::vector<int> mtw_options, ncl_options;
    int epi, pool_t;
    OutFn out_index;
    int n_stride;
    int64_t out_clip_stride;
    int out_chunk_stride;
    int lds_budget;
    int ntw;
    bool pooled = true;                // pool windows of 2x2x2 conv rows (forward) vs plain rows (input gradient)
    bool frame_tiles = false;          // first level: row groups (1, 2, 8), tiles of one frame (plan._make_plan(frame_tiles=True))
};

// plan._make_plan (pooled programs: group = 2x2x2 conv rows per pool window; plain-row programs: group = 1)
bool make_plan(const PlanSpec& sp, Plan& pl) {
    const int ntaps = (int)sp.taps.size();
    const int S = (ntaps + 1) / 2;
    std::vector<Tap> taps_p = sp.taps;
    while ((int)taps_p.size() < 2 * S) taps_p.push_back({0, 0, 0});
    int md[3] = {0, 0, 0};
    for (const auto& t : sp.taps)
        for (int k = 0; k < 3; ++k) md[k] = std::max(md[k], t[k]);
    auto ext = [&](int na, int nb, int nc, int e[3]) {
        e[0] = sp.row_stride[0] * (na - 1) + md[0] + 1;
        e[1] = sp.row_stride[1] * (nb - 1) + md[1] + 1;
        e[2] = sp.row_stride[2] * (nc - 1) + md[2] + 1;
    };
    const int RA = sp.row_dims[0], RB = sp.row_dims[1], RC = sp.row_dims[2];
    bool have = false;
    double best_cost = 0.0;
    int best_mtw = 0, best_ncl = 1, best_box[3] = {0, 0, 0};
    for (int MTW : sp.mtw_options)
        for (int ncl : sp.ncl_options) {
            const int rows_max = (sp.MW * MTW * 32) / ncl;
            const int G = sp.pooled ? 2 : 1;
            const int GA = sp.frame_tiles ? 1 : G, GB = G, GC = sp.frame_tiles ? 8 : G;
            if (rows_max < GA * GB * GC) continue;
            const int waves = (sp.NT / sp.ntw) * sp.MW;
            const int64_t dma_cap = (int64_t)waves * (MTW * sp.ntw <= 4 ? 14 : 17) * 64;
            const int64_t budget = std::min<int64_t>(sp.lds_budget, dma_cap - 64);
            // _choose_box: fewest boxes, then not-narrow patch rows, then fewest slots; first found wins ties
            bool hb = false;
            int64_t k_nbox = 0, k_slots = 0;
            int k_narrow = 0, cb[3] = {0, 0, 0};
            for (int na = GA; na < RA + GA; na += GA)
                for (int nb = GB; nb < RB + GB; nb += GB)
                    for (int nc = GC; nc < RC + GC; nc += GC) {
                        if (na > RA + GA - 1 || nb > RB + GB - 1 || nc > RC + GC - 1) continue;
                        if (na * nb * nc > rows_max) continue;
                        int e[3];
                        ext(na, nb, nc, e);
                        const int64_t slots = (int64_t)((double)((int64_t)ncl * e[0] * e[1] * e[2]) * 1.06) + 16;
                        if (slots > budget) continue;
                        const int64_t nbox = cdiv(RA, na) * cdiv(RB, nb) * cdiv(RC, nc);
                        const int narrow = nc < std::min(8, RC) ? 1 : 0;
                        const bool better = !hb || nbox < k_nbox || (nbox == k_nbox && (narrow < k_narrow ||
                                            (narrow == k_narrow && slots < k_slots)));
                        if (better) { hb = true; k_nbox = nbox; k_narrow = narrow; k_slots = slots; cb[0] = na; cb[1] = nb; cb[2] = nc; }
                    }
            if (!hb) continue;
            const int64_t nbox = cdiv(RA, cb[0]) * cdiv(RB, cb[1]) * cdiv(RC, cb[2]);
            const double cost = (double)(nbox * MTW * sp.MW) / (double)ncl;
            if (!have || cost < best_cost) {
                have = true; best_cost = cost; best_mtw = MTW; best_ncl = ncl;
                memcpy(best_box, cb, sizeof(cb));
            }
        }
    if (!have) return false;
    const int MTW = best_mtw, ncl = best_ncl, na = best_box[0], nb = best_box[1], nc = best_box[2];
    const int mt_pad = sp.MW * MTW;
    if (sp.frame_tiles && !(na == MTW && nb == 8 && nc == 8 && sp.MW == 2 && ncl == 1)) return false;
    const int64_t dma_cap_final = (int64_t)(sp.NT / sp.ntw) * sp.MW * (MTW * sp.ntw <= 4 ? 14 : 17) * 64;
    const int64_t slot_cap = std::min<int64_t>((int64_t)(sp.lds_budget * 1.12), dma_cap_final);
    int e_box[3];
    ext(na, nb, nc, e_box);
    pl = Plan();
    std::vector<std::array<int, 3>> keys;
    for (int a0 = 0; a0 < RA; a0 += na)
        for (int b0 = 0; b0 < RB; b0 += nb)
            for (int c0 = 0; c0 < RC; c0 += nc) {
                const std::array<int, 3> key = {std::min(na, RA - a0), std::min(nb, RB - b0), std::min(nc, RC - c0)};
                int ty = -1;
                for (size_t k = 0; k < keys.size(); ++k)
                    if (keys[k] == key) ty = (int)k;
                if (ty < 0) {
                    const int valid[3] = {key[0], key[1], key[2]};
                    pl.types.push_back(sp.pooled ? build_type(best_box, sp.row_stride, taps_p, e_box, mt_pad, sp.out_index, valid, ncl, slot_cap, sp.frame_tiles)
                                                 : build_type_rows(best_box, sp.row_stride, taps_p, e_box, mt_pad, sp.out_index, valid, ncl, slot_cap));
                    keys.push_back(key);
                    ty = (int)keys.size() - 1;
                }
                pl.boxes.push_back({ty, (int64_t)sp.row_stride[0] * a0 + sp.row_origin[0], (int64_t)sp.row_stride[1] * b0 + sp.row_origin[1],
                                    (int64_t)sp.row_stride[2] * c0 + sp.row_origin[2], sp.out_index(0, a0, b0, c0) - sp.out_index(0, 0, 0, 0), 0});
            }
    pl.CC = sp.CC; pl.F = sp.src_grid[0]; pl.H = sp.src_grid[1]; pl.W = sp.src_grid[2];
    pl.row_pitch4 = pl.W * 4;
    pl.chunk_stride4 = (int64_t)pl.F * pl.H * pl.W * 4;
    pl.clip_stride4 = (int64_t)sp.CC * pl.F * pl.H * pl.W * 4;
    pl.NT = sp.NT; pl.MW = sp.MW; pl.MTW = MTW; pl.S = S; pl.ncl = ncl;
    pl.epi = sp.epi; pl.pool_t = sp.pool_t; pl.relu = sp.pooled ? 1 : 0; pl.n_out = sp.n_out; pl.n_stride = sp.n_stride;
    pl.out_clip_stride = sp.out_clip_stride; pl.out_chunk_stride = sp.out_chunk_stride;
    pl.rows_total = (int64_t)pl.boxes.size() * mt_pad * 32 / ncl;
    pl.NTW = sp.ntw;
    return true;
}

int conv_out_dim(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }
int pix_row_pitch(int w) { return ((w + 8 + 7) / 8) * 8; }

// plan.plan_forward_cl: Conv3d(cin->cout) + ReLU + MaxPool(pool_t,2,2) over channels-last chunks [clip][cin/8][t][h][w][8]
bool plan_forward_cl(int cin, int cout, int t_in, int h_in, int w_in, int pool_t, bool feat_out, int lds_budget,
                     std::vector<int> mtw_options, int ntw, Plan& pl) {
    if (cin % 8 || cout % 32) return false;
    if (ntw == 2) {
        if (cout != 128) return false;
        mtw_options = {4};
    }
    const int CC = cin / 8;
    const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
    const int To = T / pool_t, Ho = OH / 2, Wo = OW / 2;
    PlanSpec sp;
    sp.src_grid[0] = t_in; sp.src_grid[1] = h_in; sp.src_grid[2] = w_in;
    sp.CC = CC;
    sp.row_dims[0] = To * pool_t; sp.row_dims[1] = Ho * 2; sp.row_dims[2] = Wo * 2;
    sp.row_origin[0] = -1; sp.row_origin[1] = -3; sp.row_origin[2] = -3;
    sp.row_stride[0] = 1; sp.row_stride[1] = 2; sp.row_stride[2] = 2;
    for (int kt = 0; kt < KT; ++kt)
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) sp.taps.push_back({kt, kh, kw});
    sp.n_out = cout; sp.NT = cout / 32; sp.MW = std::max(1, 4 / (sp.NT / ntw));
    sp.mtw_options = mtw_options;
    const int64_t rows_all = (int64_t)sp.row_dims[0] * sp.row_dims[1] * sp.row_dims[2];
    const int max_ncl = (int)std::max<int64_t>(1, (8 * 32 * sp.MW) / std::max<int64_t>(rows_all, 1));
    std::set<int> ncls = {1, max_ncl};
    for (int n : {2, 4, 8}) if (n <= max_ncl) ncls.insert(n);
    sp.ncl_options.assign(ncls.begin(), ncls.end());
    sp.pool_t = pool_t; sp.lds_budget = lds_budget; sp.ntw = ntw;
    if (feat_out) {
        const int64_t npos = (int64_t)To * Ho * Wo, feat_stride = (int64_t)cout * npos;
        sp.epi = EPI_POOL_FEAT;
        sp.out_index = [=](int ci, int a, int b, int c) { return ci * feat_stride + ((int64_t)(a / pool_t) * Ho + b / 2) * Wo + c / 2; };
        sp.n_stride = (int)npos; sp.out_clip_stride = feat_stride; sp.out_chunk_stride = 0;
    } else {
        const int64_t chunk_stride = (int64_t)To * Ho * Wo, clip_stride = (cout / 8) * chunk_stride;
        sp.epi = EPI_POOL_CL;
        sp.out_index = [=](int ci, int a, int b, int c) { return ci * clip_stride + ((int64_t)(a / pool_t) * Ho + b / 2) * Wo + c / 2; };
        sp.n_stride = 0; sp.out_clip_stride = clip_stride; sp.out_chunk_stride = (int)chunk_stride;
    }
    if (!make_plan(sp, pl)) return false;
    // weight gather index [CC][S][NT][64 lanes][8]: lane (col = lane & 31, half = lane >> 5) of K step s holds
    // W[n = nt*32 + col][c = cc*8 + j][tap 2s + half]
    const int S = pl.S, NT = pl.NT, ntaps = (int)sp.taps.size();
    pl.widx.assign((size_t)CC * S * NT * 64 * 8, -1);
    for (int cc = 0; cc < CC; ++cc)
        for (int s = 0; s < S; ++s)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int p = 2 * s + (lane >> 5);
                    if (p >= ntaps) continue;
                    const int kt = sp.taps[p][0], kh = sp.taps[p][1], kw = sp.taps[p][2];
                    const int64_t n = nt * 32 + (lane & 31);
                    for (int j = 0; j < 8; ++j) {
                        const int64_t c = cc * 8 + j;
                        pl.widx[((((size_t)cc * S + s) * NT + nt) * 64 + lane) * 8 + j] =
                            (int32_t)((((n * cin + c) * KT + kt) * KH + kh) * KW + kw);
                    }
                }
    return true;
}

// plan.plan_forward_pix: first layer over 16-bit pixel rows [clip][t*3+c][h][W+8] (vd_pix2rows); the kw-slot of output
// column ow is the 16 bytes at dword offset ow of the row
bool plan_forward_pix(int cout, int t_in, int h_in, int w_in, int lds_budget, int ntw, Plan& pl) {
    const int cin = 3;
    const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
    const int Ho = OH / 2, Wo = OW / 2;
    PlanSpec sp;
    sp.src_grid[0] = t_in * cin; sp.src_grid[1] = h_in; sp.src_grid[2] = OW;
    sp.CC = 1;
    sp.row_dims[0] = T; sp.row_dims[1] = Ho * 2; sp.row_dims[2] = Wo * 2;
    sp.row_origin[0] = -cin; sp.row_origin[1] = -3; sp.row_origin[2] = 0;
    sp.row_stride[0] = cin; sp.row_stride[1] = 2; sp.row_stride[2] = 1;
    sp.n_out = cout; sp.NT = cout / 32; sp.MW = std::max(1, 4 / (sp.NT / ntw));
    sp.mtw_options = {4};
    if (ntw == 2) {
        if (sp.NT != 2) return false;
        sp.mtw_options = {2};
    }
    sp.ncl_options = {1};
    sp.epi = EPI_POOL_CL; sp.pool_t = 1; sp.lds_budget = lds_budget; sp.ntw = ntw;
    const int64_t chunk_stride = (int64_t)T * Ho * Wo, clip_stride = (cout / 8) * chunk_stride;
    sp.out_index = [=](int ci, int a, int b, int c) { return ci * clip_stride + ((int64_t)a * Ho + b / 2) * Wo + c / 2; };
    sp.n_stride = 0; sp.out_clip_stride = clip_stride; sp.out_chunk_stride = (int)chunk_stride;
    bool done = false;
    const char* ft = getenv("VD_L0_FRAME_TILES");
    if ((ft == nullptr || strcmp(ft, "1") == 0) && ntw == 1 && sp.NT == 2 && Wo % 4 == 0) {
        // FRAME TILES (plan.plan_forward_pix): K order (tap pair j, kt) over the first 20 (c, kh) taps of a kt plane, then the
        // three left-over taps
        int q[21][2];
        for (int c = 0; c < cin; ++c)
            for (int kh = 0; kh < KH; ++kh) { q[c * KH + kh][0] = c; q[c * KH + kh][1] = kh; }
        for (int j = 0; j < 10; ++j)
            for (int kt = 0; kt < KT; ++kt)
                for (int e = 0; e < 2; ++e) sp.taps.push_back({kt * cin + q[2 * j + e][0], q[2 * j + e][1], 0});
        for (int kt = 0; kt < KT; ++kt) sp.taps.push_back({kt * cin + q[20][0], q[20][1], 0});
        sp.frame_tiles = true;
        if (make_plan(sp, pl)) {
            // what the frame-sharing kernel relies on (plan.plan_forward_pix checks the same; a violation falls back to the frame-pair layout):
            // tile i of a wave row = tile 0 + i frames, tap 3 j + kt = tap 3 j + kt frames
            bool ok = true;
            for (const BoxType& t : pl.types) {
                const int fs = cin * t.pitch_f * SLOT_BYTES;
                if ((int)t.a_off.size() < sp.MW * 4 * 32 || (int)t.tap_off.size() < 2 * 30) { ok = false; break; }
                for (int w = 0; w < sp.MW && ok; ++w)
                    for (int i = 1; i < 4 && ok; ++i)
                        for (int r = 0; r < 32; ++r)
                            if (t.a_off[(w * 4 + i) * 32 + r] - t.a_off[(w * 4) * 32 + r] != i * fs) { ok = false; break; }
                for (int j = 0; j < 10 && ok; ++j)
                    for (int kt = 1; kt < KT && ok; ++kt)
                        for (int e = 0; e < 2; ++e)
                            if (t.tap_off[2 * (3 * j + kt) + e] - t.tap_off[2 * (3 * j) + e] != kt * fs) { ok = false; break; }
            }
            if (ok) {
                pl.out_t_stride = FRAME_TILE_OUT_STEP;
                pl.pair_flip = FRAME_TILE_FLIP | (pl.types[0].pitch_h << 8) | (pl.types[0].pitch_f << 16);
                done = true;
            }
        }
    }
    if (!done) {
        if (T % 2) return false;
        sp.frame_tiles = false;
        sp.taps.clear();
        for (int kt = 0; kt < KT; ++kt)
            for (int c = 0; c < cin; ++c)
                for (int kh = 0; kh < KH; ++kh) sp.taps.push_back({kt * cin + c, kh, 0});
        if (!make_plan(sp, pl)) return false;
        pl.out_t_stride = Ho * Wo;
    }
    const int rowp = pix_row_pitch(w_in);
    pl.w_step4 = 1; pl.row_pitch4 = rowp / 2;
    pl.chunk_stride4 = pl.clip_stride4 = (int64_t)t_in * cin * h_in * (rowp / 2);
    const int S = pl.S, NT = pl.NT, ntaps = (int)sp.taps.size();
    pl.widx.assign((size_t)S * NT * 64 * 8, -1);
    for (int s = 0; s < S; ++s)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int p = 2 * s + (lane >> 5);
                if (p >= ntaps) continue;
                const int fc = sp.taps[p][0], kh = sp.taps[p][1];
                const int kt = fc / cin, c = fc % cin;
                const int64_t n = nt * 32 + (lane & 31);
                for (int kw = 0; kw < 7; ++kw)
                    pl.widx[(((size_t)s * NT + nt) * 64 + lane) * 8 + kw] = (int32_t)((((n * cin + c) * KT + kt) * KH + kh) * KW + kw);
            }
    return true;
}

// plan.plan_dgrad: input gradient of Conv3d(cin->cout) for the input positions (t, 2b+ph, 2c+pw); source = dense dy on the
// conv grid (channels-last chunks of cout); dx[t,h,w,ci] = sum dy[t+1-kt, (h+3-kh)/2, (w+3-kw)/2, n] * W[n,ci,kt,kh,kw]
bool plan_dgrad(int cin, int cout, int t_in, int h_in, int w_in, int ph, int pw, int lds_budget, std::vector<int> mtw_options, Plan& pl) {
    if (cout % 8) return false;
    const int CC = cout / 8;
    const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
    std::vector<Tap> tap_k;
    for (int kt = 0; kt < KT; ++kt)
        for (int kh = 0; kh < KH; ++kh) {
            if ((ph + 3 - kh) % 2 != 0) continue;
            for (int kw = 0; kw < KW; ++kw)
                if ((pw + 3 - kw) % 2 == 0) tap_k.push_back({kt, kh, kw});
        }
    PlanSpec sp;
    for (const auto& k : tap_k) {
        // python floor division: (ph + 3 - kh) // 2 + 1 with possibly negative numerators
        auto fdiv2 = [](int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); };
        sp.taps.push_back({2 - k[0], fdiv2(ph + 3 - k[1]) + 1, fdiv2(pw + 3 - k[2]) + 1});
    }
    sp.src_grid[0] = T; sp.src_grid[1] = OH; sp.src_grid[2] = OW;
    sp.CC = CC;
    sp.row_dims[0] = t_in; sp.row_dims[1] = (h_in - ph + 1) / 2; sp.row_dims[2] = (w_in - pw + 1) / 2;
    sp.row_origin[0] = sp.row_origin[1] = sp.row_origin[2] = -1;
    sp.row_stride[0] = sp.row_stride[1] = sp.row_stride[2] = 1;
    const int n_pad = ((cin + 31) / 32) * 32;
    sp.n_out = cin; sp.NT = n_pad / 32; sp.MW = std::max(1, 4 / sp.NT);
    sp.mtw_options = mtw_options;
    const int64_t rows_all = (int64_t)sp.row_dims[0] * sp.row_dims[1] * sp.row_dims[2];
    const int max_ncl = (int)std::max<int64_t>(1, (8 * 32 * sp.MW) / std::max<int64_t>(rows_all, 1));
    std::set<int> ncls = {1, max_ncl};
    for (int n : {2, 4, 8}) if (n <= max_ncl) ncls.insert(n);
    sp.ncl_options.assign(ncls.begin(), ncls.end());
    sp.epi = EPI_ROWS; sp.pool_t = 0; sp.lds_budget = lds_budget; sp.ntw = 1; sp.pooled = false;
    const int64_t clip_stride = (int64_t)t_in * h_in * w_in * cin;       // fp32 [t][h][w][cin]
    sp.out_index = [=](int ci, int a, int b, int c) { return ci * clip_stride + (((int64_t)a * h_in + (2 * b + ph)) * w_in + 2 * c + pw) * cin; };
    sp.n_stride = 1; sp.out_clip_stride = clip_stride; sp.out_chunk_stride = 0;
    if (!make_plan(sp, pl)) return false;
    const int S = pl.S, NT = pl.NT, ntaps = (int)tap_k.size();
    pl.widx.assign((size_t)CC * S * NT * 64 * 8, -1);
    for (int cc = 0; cc < CC; ++cc)
        for (int s = 0; s < S; ++s)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int p = 2 * s + (lane >> 5);
                    if (p >= ntaps) continue;
                    const int64_t ci = nt * 32 + (lane & 31);
                    if (ci >= cin) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int64_t n = cc * 8 + j;
                        pl.widx[((((size_t)cc * S + s) * NT + nt) * 64 + lane) * 8 + j] =
                            (int32_t)((((n * cin + ci) * KT + tap_k[p][0]) * KH + tap_k[p][1]) * KW + tap_k[p][2]);
                    }
                }
    return true;
}

// plan.plan_dgrad_pix: input gradient of the FIRST layer with the stride-2 parity classes merged into N
// (one output row = the 2 x bw pixel block (t, 2b..2b+1, bw*c..bw*c+bw-1), columns n = (ci*2 + ph)*bw + pw;
//  bw = 2: N = 12, K = 48 taps x cout;  bw = 4: N = 24, K = 60 taps x cout, half the rows: 0.625 of the MFMA work)
bool plan_dgrad_pix(int cin, int cout, int t_in, int h_in, int w_in, int lds_budget, Plan& pl, int bw = 2,
                    const std::vector<int>& mtw_options = {7, 8}) {
    if ((bw != 2 && bw != 4) || cin * 2 * bw > 32 || cout % 8 || h_in % 2 || w_in % bw) return false;
    const int CC = cout / 8;
    const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
    PlanSpec sp;
    for (int dt = 0; dt < 3; ++dt)
        for (int dh = 0; dh < 4; ++dh)
            for (int dw = 0; dw < bw / 2 + 3; ++dw) sp.taps.push_back({dt, dh, dw});
    sp.src_grid[0] = T; sp.src_grid[1] = OH; sp.src_grid[2] = OW;
    sp.CC = CC;
    sp.row_dims[0] = t_in; sp.row_dims[1] = h_in / 2; sp.row_dims[2] = w_in / bw;
    sp.row_origin[0] = sp.row_origin[1] = sp.row_origin[2] = -1;
    sp.row_stride[0] = sp.row_stride[1] = 1; sp.row_stride[2] = bw / 2;
    sp.n_out = cin * 2 * bw; sp.NT = 1; sp.MW = 4;
    sp.mtw_options = mtw_options;
    sp.ncl_options = {1};
    sp.epi = EPI_ROWS; sp.pool_t = 0; sp.lds_budget = lds_budget; sp.ntw = 1; sp.pooled = false;
    const int64_t clip_stride = (int64_t)t_in * cin * h_in * w_in;
    sp.out_index = [=](int ci, int a, int b, int c) { return ci * clip_stride + ((int64_t)a * cin * h_in + 2 * b) * w_in + bw * c; };
    sp.n_stride = 0; sp.out_clip_stride = clip_stride; sp.out_chunk_stride = 0;
    if (!make_plan(sp, pl)) return false;
    const int S = pl.S, ntaps = (int)sp.taps.size();
    pl.widx.assign((size_t)CC * S * 64 * 8, -1);
    for (int cc = 0; cc < CC; ++cc)
        for (int s = 0; s < S; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int p = 2 * s + (lane >> 5);
                if (p >= ntaps) continue;
                const int dt = sp.taps[p][0], dh = sp.taps[p][1], dw = sp.taps[p][2];
                const int kt = 2 - dt, n = lane & 31;
                const int ci = n / (2 * bw), ph = (n / bw) % 2, pw = n % bw;
                const int kh = ph + 5 - 2 * dh, kw = pw + 5 - 2 * dw;
                if (ci >= cin || kh < 0 || kh >= KH || kw < 0 || kw >= KW) continue;
                for (int j = 0; j < 8; ++j) {
                    const int64_t nn = cc * 8 + j;
                    pl.widx[(((size_t)cc * S + s) * 64 + lane) * 8 + j] = (int32_t)((((nn * cin + ci) * KT + kt) * KH + kh) * KW + kw);
                }
            }
    pl.col_off.assign(32, 0);
    for (int n = 0; n < cin * 2 * bw; ++n) pl.col_off[n] = (n / (2 * bw)) * h_in * w_in + ((n / bw) % 2) * w_in + (n % bw);
    return true;
}

// plan.bwd0_block_w: 2 x 4 pixel blocks where the width allows (VD_BWD0_WIDE=0: the 2 x 2 blocks of rounds 1-3, for A/B runs)
int bwd0_block_w(int w_in) {
    const char* e = getenv("VD_BWD0_WIDE");
    return (w_in % 4 == 0 && !(e != nullptr && e[0] == '0')) ? 4 : 2;
}

// input geometry (cin, t, h, w) of ConvNet3D level `layer`
void layer_input(int layer, int frames, int height, int width, int& cin, int& t, int& h, int& w) {
    const int widths[3] = {64, 128, 128}, pools_t[3] = {1, 2, 2};
    cin = 3; t = frames; h = height; w = width;
    for (int li = 0; li < layer; ++li) {
        const int T = conv_out_dim(t, KT, 1, 1), OH = conv_out_dim(h, KH, 2, 3), OW = conv_out_dim(w, KW, 2, 3);
        cin = widths[li]; t = T / pools_t[li]; h = OH / 2; w = OW / 2;
    }
}

// plan.latency_variant's acceptance rule: more workgroups, and either (almost) no more padded rows than the base program or --
// a TINY launch, fewer workgroups than a quarter of the chip's 512 slots -- fewer M tiles per wave whatever it pads
bool latency_better(const Plan& alt, const Plan& best, const Plan& base, int batch_hint) {
    if (alt.grid(batch_hint) <= best.grid(batch_hint)) return false;
    const char* e = getenv("VD_TINY_GRID");
    const int tiny_grid = e != nullptr ? atoi(e) : 128;
    const bool tiny = best.grid(batch_hint) < tiny_grid && alt.MTW < best.MTW && alt.grid(batch_hint) <= 512;
    return (double)alt.rows_total <= (double)base.rows_total * 1.05 || tiny;
}

// the input-gradient programs of `layer` as plan.plan_network builds them: layer 0 -> one merged program (even H, W);
// layers 1, 2 -> one program per stride-2 parity class (index = ph * 2 + pw), latency-oriented variant for small batches
bool plan_dgrad_layer(int layer, int cls, int frames, int height, int width, int batch_hint, Plan& pl) {
    int cin, t, h, w;
    layer_input(layer, frames, height, width, cin, t, h, w);
    const int widths[3] = {64, 128, 128};
    const int cout = widths[layer], lds_budget = 3700;
    if (layer == 0) return cls == 0 && h % 2 == 0 && w % 2 == 0 && plan_dgrad_pix(cin, cout, t, h, w, lds_budget, pl, bwd0_block_w(w));   // (odd clip sizes: Python planner only)
    const int nph = std::min(2, h), npw = std::min(2, w);
    if (cls < 0 || cls >= nph * npw) return false;
    const int ph = cls / npw, pw = cls % npw;
    if (!plan_dgrad(cin, cout, t, h, w, ph, pw, lds_budget, {7, 8, 4, 2}, pl)) return false;
    if (batch_hint > 0 && pl.grid(batch_hint) < 512) {
        const Plan base = pl;
        const std::vector<std::vector<int>> tries = {{2, 4, 7, 8}, {4}, {2}};
        for (const auto& opts : tries) {
            Plan alt;
            if (!plan_dgrad(cin, cout, t, h, w, ph, pw, lds_budget, opts, alt)) continue;
            if (latency_better(alt, pl, base, batch_hint)) pl = alt;
        }
    }
    return true;
}

// the forward program of `layer` as plan.plan_network + engine.EmbedEngine choose it for operand precision `prec`
bool plan_layer(int layer, int frames, int height, int width, int prec, int batch_hint, Plan& pl) {
    const bool x3 = (prec == VD_PREC_BF16X3 || prec == VD_PREC_F16X3);
    const int lds_budget = 3700;
    int cin = 3, t = frames, h = height, w = width;
    const int widths[3] = {64, 128, 128}, pools_t[3] = {1, 2, 2};
    for (int li = 0; li < layer; ++li) {
        const int T = conv_out_dim(t, KT, 1, 1), OH = conv_out_dim(h, KH, 2, 3), OW = conv_out_dim(w, KW, 2, 3);
        cin = widths[li]; t = T / pools_t[li]; h = OH / 2; w = OW / 2;
    }
    if (layer == 0) return plan_forward_pix(widths[0], t, h, w, lds_budget, 1, pl);   // (x1: the register-resident-B kernel's layout; x3: one N tile)
    const int cout = widths[layer], pt = pools_t[layer];
    const bool feat = (layer == 2);
    const std::vector<int> dflt = {7, 8, 4, 2};
    if (!plan_forward_cl(cin, cout, t, h, w, pt, feat, lds_budget, dflt, 1, pl)) return false;
    Plan pl2;
    if (cout == 128 && plan_forward_cl(cin, cout, t, h, w, pt, feat, lds_budget, dflt, 2, pl2) && pl2.rows_total <= pl.rows_total) {
        pl = pl2;
    } else if (!x3 && pl.NT == 4 && pl.MW == 1 && pl.MTW == 7) {
        bool all7 = true;
        for (const auto& ty : pl.types) all7 = all7 && ty.mt == 7;
        if (all7) { pl.MW = 2; pl.MTW = 3; pl.NTW = 2; }      // balanced 7-tile layout (kernel template BAL)
    }
    // latency_variant: a launch that starts fewer workgroups than the chip has slots prefers more, shorter workgroups
    if (batch_hint > 0 && pl.grid(batch_hint) < 512) {
        const Plan base = pl;
        const std::vector<std::vector<int>> tries = {{2, 4, 7, 8}, {4}, {2}};
        for (const auto& opts : tries) {
            Plan alt;
            if (!plan_forward_cl(cin, cout, t, h, w, pt, feat, lds_budget, opts, 1, alt)) continue;
            if (latency_better(alt, pl, base, batch_hint)) pl = alt;
        }
    }
    return true;
}

// plan._plan_wgrad: weight gradient of Conv3d(cin->cout, k(3,7,7), s(1,2,2), p(1,3,3)) as a tile program with the operand
// roles rotated -- a program "clip" is an input channel, its 147 output rows the taps, K runs over (position of a block) x
// (8 clips per slot); boxes of positions accumulate, fp32 atomics, into [copy][cin][tap][cout] scratch.
void plan_wgrad_block(int cin, int cout, int t_in, int h_in, int w_in, int nclips, int lds_budget, int bt0, int bh0, int bw0, Plan& pl) {
    const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
    int nt = std::min(bt0, T), noh = std::min(bh0, OH), now = std::min(bw0, OW);
    if ((nt * noh * now) % 2) now += 1;
    const int CCb = (nclips + 7) / 8, NT = cout / 32, MTW = 5;
    const int pf = nt + 2, ph = 2 * (noh - 1) + 7, pw = 2 * (now - 1) + 7;
    std::vector<std::array<int, 3>> positions;
    for (int dt = 0; dt < nt; ++dt) for (int doh = 0; doh < noh; ++doh) for (int dow = 0; dow < now; ++dow) positions.push_back({dt, doh, dow});
    const int S = (int)positions.size() / 2;
    const int ntaps = KT * KH * KW;
    bool have = false;
    double best_cyc = 0.0; int best_pc = 0;
    BoxType best;
    for (int dph = 0; dph < 16; ++dph) {
        const int pitch_h = pw + dph;
        for (int dpf = 0; dpf < 16; ++dpf) {
            const int pitch_f = ph * pitch_h + dpf, pitch_c = pf * pitch_f;
            if ((double)pitch_c > lds_budget * 1.12 && have) continue;
            std::vector<int64_t> a_off((size_t)MTW * 32, 0);
            int r = 0;
            for (int kt = 0; kt < KT; ++kt) for (int kh = 0; kh < KH; ++kh) for (int kw = 0; kw < KW; ++kw) a_off[r++] = (int64_t)kt * pitch_f + kh * pitch_h + kw;
            double cyc = 0.0;
            for (int t = 0; t < MTW; ++t) cyc += conflict_cycles(a_off.data() + t * 32);
            cyc /= MTW;
            const double rc = std::nearbyint(cyc * 1000.0) / 1000.0;          // Python round(cyc, 3) (half-even, like nearbyint's default mode)
            if (!have || rc < best_cyc || (rc == best_cyc && pitch_c < best_pc)) {
                have = true; best_cyc = rc; best_pc = pitch_c;
                best = BoxType();
                best.pf = pf; best.ph = ph; best.pw = pw; best.pitch_h = pitch_h; best.pitch_f = pitch_f; best.pitch_c = pitch_c; best.mt = MTW;
                best.cyc = cyc;
                best.a_off.resize((size_t)MTW * 32); best.out.assign((size_t)MTW * 32, -1);
                for (int i = 0; i < MTW * 32; ++i) best.a_off[i] = (int32_t)(a_off[i] * 16);
                for (int i = 0; i < ntaps; ++i) best.out[i] = i * cout;
                best.tap_off.resize(positions.size());
                for (size_t i = 0; i < positions.size(); ++i)
                    best.tap_off[i] = (int32_t)(((int64_t)positions[i][0] * pitch_f + 2 * positions[i][1] * pitch_h + 2 * positions[i][2]) * 16);
            }
            if (cyc <= 4.0 + 1e-9) break;
        }
        if (best_cyc <= 4.0 + 1e-9) break;
    }
    pl = Plan();
    pl.types.push_back(best);
    for (int t0 = 0; t0 < T; t0 += nt) for (int oh0 = 0; oh0 < OH; oh0 += noh) for (int ow0 = 0; ow0 < OW; ow0 += now)
        pl.boxes.push_back({0, (int64_t)t0 - 1, (int64_t)2 * oh0 - 3, (int64_t)2 * ow0 - 3, 0, 0});
    // (ordered mode, vd_set_deterministic: every box gets its OWN copy -- an atomic add onto a zeroed float with a single
    //  contributor is exact, so the only summation left is vd_replica_sum's fixed-order fold over the copies)
    const int replicas = vd_get_deterministic() ? pl.nbox() : std::max(1, std::min(16, pl.nbox() / 28));
    for (int bi = 0; bi < pl.nbox(); ++bi) pl.boxes[bi][5] = bi % replicas;
    pl.CC = CCb; pl.F = t_in; pl.H = h_in; pl.W = w_in; pl.row_pitch4 = w_in * 4; pl.w_step4 = 4;
    pl.chunk_stride4 = (int64_t)t_in * h_in * w_in * 4; pl.clip_stride4 = (int64_t)CCb * t_in * h_in * w_in * 4;
    pl.NT = NT; pl.MW = 1; pl.MTW = MTW; pl.S = S; pl.ncl = 1; pl.NTW = 1;
    pl.epi = EPI_ROWS; pl.pool_t = 0; pl.relu = 0; pl.n_out = cout; pl.n_stride = 1;
    pl.out_clip_stride = (int64_t)ntaps * cout; pl.out_chunk_stride = 0; pl.out_t_stride = 0;
    pl.atomic = 1; pl.w_box_stride = (int64_t)CCb * S * NT * 64 * 8;
    pl.rows_total = (int64_t)pl.nbox() * MTW * 32;
}

// plan.plan_wgrad: the block of positions with the fewest K steps (incl. padding of partial blocks) per resident wave
bool plan_wgrad(int cin, int cout, int t_in, int h_in, int w_in, int nclips, int planes, Plan& out, int block[3], int& replicas) {
    static const int BLOCKS[12][3] = {{8, 4, 4}, {4, 4, 14}, {8, 2, 14}, {4, 7, 7}, {2, 7, 14}, {4, 2, 14}, {2, 4, 14}, {4, 4, 4},
                                      {2, 2, 14}, {1, 4, 14}, {2, 4, 4}, {1, 2, 14}};
    const int lds_budget = 3700, wg_waves = cout / 32, want = nclips > 8 ? 6 : 4;
    bool have = false;
    double best_score = 0.0; int best_nbox = 0, best_S = 0;
    for (const auto& b : BLOCKS) {
        Plan pl;
        plan_wgrad_block(cin, cout, t_in, h_in, w_in, nclips, lds_budget, b[0], b[1], b[2], pl);
        const BoxType& bt = pl.types[0];
        const int64_t groups = cdiv(bt.pitch_c, 64);
        if (bt.pitch_c > lds_budget || groups > (int64_t)wg_waves * 17) continue;
        const int64_t lds = (int64_t)planes * bt.pitch_c * 16 + 2048;
        const int64_t resident = std::min<int64_t>(std::min<int64_t>(8, (160 * 1024 / lds) * wg_waves), (int64_t)cin * pl.nbox() * wg_waves / 256);
        const double score = (double)((int64_t)pl.nbox() * pl.S) / (double)std::max<int64_t>(1, std::min<int64_t>(resident, want));
        const bool better = !have || score < best_score || (score == best_score && (pl.nbox() < best_nbox || (pl.nbox() == best_nbox && -pl.S < -best_S)));
        if (better) {
            have = true; best_score = score; best_nbox = pl.nbox(); best_S = pl.S;
            out = pl;
            const int T = conv_out_dim(t_in, KT, 1, 1), OH = conv_out_dim(h_in, KH, 2, 3), OW = conv_out_dim(w_in, KW, 2, 3);
            block[0] = std::min(b[0], T); block[1] = std::min(b[1], OH); block[2] = std::min(b[2], OW);
            if ((block[0] * block[1] * block[2]) % 2) block[2] += 1;
        }
    }
    if (!have) return false;
    replicas = 1;
    for (int bi = 0; bi < out.nbox(); ++bi) replicas = std::max(replicas, (int)out.boxes[bi][5] + 1);
    return true;
}

// plan.export_program: 40 int64 header words + int32 arrays type_desc | tables | boxes | gather | widx | col_off
// (box walks of the first-level kernels: the grid is this many generations of resident workgroups -- see plan.export_program)
#define VD_BOX_WALK_GENERATIONS 16      // plan.BOX_WALK_GENERATIONS
std::vector<uint8_t> export_program(const Plan& pl, int persist) {
    std::vector<int32_t> desc, tables;
    int64_t pos = 0;
    for (const auto& t : pl.types) {
        const int64_t a_ofs = pos; pos += (int64_t)t.a_off.size();
        const int64_t o_ofs = pos; pos += (int64_t)t.out.size();
        const int64_t t_ofs = pos; pos += (int64_t)t.tap_off.size();
        const int32_t row[16] = {t.pf, t.ph, t.pw, t.pitch_h, t.pitch_f, t.pitch_c, t.mt, (int32_t)a_ofs, (int32_t)o_ofs, (int32_t)t_ofs,
                                 0, 0, 0, 0, 0, 0};
        desc.insert(desc.end(), row, row + 16);
        tables.insert(tables.end(), t.a_off.begin(), t.a_off.end());
        tables.insert(tables.end(), t.out.begin(), t.out.end());
        tables.insert(tables.end(), t.tap_off.begin(), t.tap_off.end());
    }
    const int nbox = pl.nbox();
    std::vector<int32_t> boxes((size_t)nbox * 8, 0);
    for (int bi = 0; bi < nbox; ++bi) {
        const int ty = (int)pl.boxes[bi][0];
        boxes[(size_t)bi * 8 + 0] = desc[(size_t)ty * 16 + 7];
        boxes[(size_t)bi * 8 + 1] = desc[(size_t)ty * 16 + 8];
        boxes[(size_t)bi * 8 + 2] = desc[(size_t)ty * 16 + 9];
        boxes[(size_t)bi * 8 + 3] = (int32_t)pl.boxes[bi][4];
        boxes[(size_t)bi * 8 + 4] = ty;
        boxes[(size_t)bi * 8 + 5] = (int32_t)pl.boxes[bi][5];      // copy of the accumulation target (weight-gradient programs)
        // patch origin in source slot coordinates: (f0 << 16) | (h0 & 0xffff), w0 -- read by the first-level kernel that builds
        // its patch from aligned row loads (conv0_breg3_kernel)
        boxes[(size_t)bi * 8 + 6] = (int32_t)(((uint32_t)(int32_t)pl.boxes[bi][1] << 16) | ((uint32_t)(int32_t)pl.boxes[bi][2] & 0xFFFFu));
        boxes[(size_t)bi * 8 + 7] = (int32_t)pl.boxes[bi][3];
    }
    // gather table: for every LDS slot of a box's patch the source slot it is filled from (dword offset inside the
    // (clip, chunk) block, box-local clip index in bits 24..30); -1 = zero fill
    const int64_t gstride = cdiv(pl.lds_slots() + 1, 64) * 64;
    std::vector<int32_t> gt((size_t)nbox * gstride, -1);
    for (int bi = 0; bi < nbox; ++bi) {
        const BoxType& t = pl.types[(size_t)pl.boxes[bi][0]];
        const int64_t f0 = pl.boxes[bi][1], h0 = pl.boxes[bi][2], w0 = pl.boxes[bi][3];
        const int64_t n = (int64_t)pl.ncl * t.pitch_c;
        for (int64_t idx = 0; idx < n; ++idx) {
            const int64_t ci = idx / t.pitch_c, r1 = idx % t.pitch_c;
            const int64_t f = r1 / t.pitch_f, r2 = r1 % t.pitch_f;
            const int64_t h = r2 / t.pitch_h, w = r2 % t.pitch_h;
            const int64_t sf = f0 + f, sh = h0 + h, sw = w0 + w;
            const bool ok = f < t.pf && h < t.ph && w < t.pw && sf >= 0 && sf < pl.F && sh >= 0 && sh < pl.H && sw >= 0 && sw < pl.W;
            if (ok) gt[(size_t)bi * gstride + idx] = (int32_t)(((sf * pl.H + sh) * pl.row_pitch4 + sw * pl.w_step4) | (ci << 24));
        }
    }
    int mt_max = 0;
    for (const auto& t : pl.types) mt_max = std::max(mt_max, t.mt);
    int64_t h[40];
    memset(h, 0, sizeof(h));
    memcpy(&h[0], "VDPROG01", 8);
    const int64_t head[27] = {pl.CC, pl.S, pl.NT, pl.MW, pl.MTW, pl.NTW, pl.epi, pl.pool_t, pl.relu, pl.n_out, pl.n_stride,
                              pl.out_clip_stride, pl.out_chunk_stride, pl.out_t_stride, gstride * 16, (int64_t)pl.types.size(),
                              desc[7], desc[8], desc[9], pl.atomic, pl.w_box_stride, pl.ncl, nbox, gstride, pl.clip_stride4, pl.chunk_stride4,
                              (pl.NTW == 2 && mt_max < pl.MW * pl.MTW) ? mt_max : 0};
    memcpy(&h[1], head, sizeof(head));
    const int64_t sizes[7] = {(int64_t)desc.size(), (int64_t)tables.size(), (int64_t)boxes.size(), (int64_t)gt.size(),
                              (int64_t)pl.widx.size(), (int64_t)pl.col_off.size(), persist};
    memcpy(&h[28], sizes, sizeof(sizes));
    h[37] = pl.pair_flip;
    {   // words 35, 36: planes / rows per plane of the source clip when the patch can be built from aligned 16-byte row loads
        // (plan.ConvPlan.row_source: first-level programs over pixel rows, 2 x 2 waves of 4 M tiles, 8 output columns per box)
        bool ok = pl.w_step4 == 1 && pl.CC == 1 && pl.ncl == 1 && pl.NTW <= 1 && pl.types.size() == 1 && pl.NT == 2 && pl.MW == 2 &&
                  pl.MTW == 4 && pl.S == 32 && pl.types[0].pw == 8 && pl.row_pitch4 % 4 == 0 && 2 * pl.types[0].pf * pl.types[0].ph <= 768;
        for (int bi = 0; ok && bi < nbox; ++bi)
            ok = pl.boxes[bi][3] % 4 == 0 && pl.boxes[bi][3] >= 0 && std::llabs(pl.boxes[bi][1]) < 32768 && std::llabs(pl.boxes[bi][2]) < 32768;
        h[35] = ok ? pl.F : 0;
        h[36] = ok ? pl.H : 0;
    }
    std::vector<uint8_t> blob(sizeof(h));
    memcpy(blob.data(), h, sizeof(h));
    auto append = [&](const std::vector<int32_t>& v) {
        const size_t o = blob.size();
        blob.resize(o + v.size() * sizeof(int32_t));
        if (!v.empty()) memcpy(blob.data() + o, v.data(), v.size() * sizeof(int32_t));
    };
    append(desc); append(tables); append(boxes); append(gt); append(pl.widx); append(pl.col_off);
    return blob;
}

}  // namespace

extern "C" int vd_program_build(int layer, int frames, int height, int width, int prec, int batch_hint, void** blob, int64_t* nbytes) {
    if (blob == nullptr || nbytes == nullptr || layer < 0 || layer > 2 || prec < 0 || prec > 3) return -1;
    if (frames < 2 || height < 16 || width < 16) return -2;
    Plan pl;
    if (!plan_layer(layer, frames, height, width, prec, batch_hint, pl)) return -3;
    const std::vector<uint8_t> b = export_program(pl, VD_BOX_WALK_GENERATIONS);
    void* out = malloc(b.size());
    if (out == nullptr) return -5;
    memcpy(out, b.data(), b.size());
    *blob = out;
    *nbytes = (int64_t)b.size();
    return 0;
}

extern "C" int vd_program_build_dgrad(int layer, int parity_class, int frames, int height, int width, int batch_hint, void** blob,
                                      int64_t* nbytes) {
    if (blob == nullptr || nbytes == nullptr || layer < 0 || layer > 2) return -1;
    if (frames < 2 || height < 16 || width < 16) return -2;
    Plan pl;
    if (!plan_dgrad_layer(layer, parity_class, frames, height, width, batch_hint, pl)) return -3;
    const std::vector<uint8_t> b = export_program(pl, VD_BOX_WALK_GENERATIONS);
    void* out = malloc(b.size());
    if (out == nullptr) return -5;
    memcpy(out, b.data(), b.size());
    *blob = out;
    *nbytes = (int64_t)b.size();
    return 0;
}

extern "C" int vd_program_build_wgrad(int layer, int frames, int height, int width, int nclips, int planes, void** blob,
                                      int64_t* nbytes, int* block3, int* replicas) {
    if (blob == nullptr || nbytes == nullptr || block3 == nullptr || replicas == nullptr || layer < 0 || layer > 2) return -1;
    if (frames < 2 || height < 16 || width < 16 || nclips < 1 || planes < 1 || planes > 2) return -2;
    int cin, t, h, w;
    layer_input(layer, frames, height, width, cin, t, h, w);
    const int cout = layer == 0 ? 64 : 128;
    Plan pl;
    if (!plan_wgrad(cin, cout, t, h, w, nclips, planes, pl, block3, *replicas)) return -3;
    const std::vector<uint8_t> b = export_program(pl, VD_BOX_WALK_GENERATIONS);
    void* out = malloc(b.size());
    if (out == nullptr) return -5;
    memcpy(out, b.data(), b.size());
    *blob = out;
    *nbytes = (int64_t)b.size();
    return 0;
}

extern "C" void vd_blob_free(void* blob) { free(blob); }

// Process-wide switch of the accumulation order (include/vd_hip.h): initial value from the environment (VD_DETERMINISTIC=1).
namespace {
std::atomic<int> g_deterministic{-1};
}
extern "C" int vd_get_deterministic(void) {
    int v = g_deterministic.load();
    if (v < 0) {
        const char* e = getenv("VD_DETERMINISTIC");
        v = (e != nullptr && e[0] == '1') ? 1 : 0;
        g_deterministic.store(v);
    }
    return v;
}
extern "C" int vd_set_deterministic(int on) {
    const int prev = vd_get_deterministic();
    g_deterministic.store(on ? 1 : 0);
    return prev;
}
