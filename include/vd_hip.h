/* C ABI of libvd_hip.so -- the MI355X (gfx950) kernels behind the DM / gradient-matching
 * hot path of yuz1wan/video_distillation.
 *
 * The reference has no FFI: its boundary is Python (`get_network().embed`, `match_loss`,
 * `Conv3DNet`, the SGD step on the synthetic pixels).  Each entry point below names the
 * reference call it stands in for (paths under the reference repository).  All entry
 * points take raw device pointers, sizes and a hipStream_t (passed as void*), enqueue work
 * on that stream without synchronising, never allocate or free, and return 0 on success or
 * a non-zero hipError_t / negative argument-error code.
 */
#ifndef VD_HIP_H
#define VD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD_ABI_VERSION 5

/* operand precision of the MFMA contraction (accumulation is always fp32) */
#define VD_PREC_BF16   0   /* bf16 operands, one MFMA per product                       */
#define VD_PREC_F16    1   /* fp16 operands, one MFMA per product                       */
#define VD_PREC_BF16X3 2   /* hi/lo split bf16 operands, three MFMAs (fp32-class error) */
#define VD_PREC_F16X3  3   /* hi/lo split fp16 operands, three MFMAs                    */
/* fp16 pairs carry 22 bits only while the LOW part stays a normal fp16 number, i.e. for |v| >= 2^-2; below that the pair is a
 * fixed-point number with an absolute step of 2^-25.  Conv3d weights of this network are 0.004 .. 0.05 in magnitude (PyTorch's
 * default initialisation of networks.py:799: bound 1 / sqrt(fan_in)), where an unscaled pair is exact to 3e-6 .. 6e-7 of the weight
 * -- no better than a bf16 pair, and the SAME error for every clip of a batch.  The weight-packing entry points therefore
 * store W x 2^VD_F16X3_WSHIFT in VD_PREC_F16X3 operands (exact to 5e-8 of the weight, fp32's own rounding), and VD_PREC_F16X3 tile
 * programs whose B operand is packed weights (w_box_stride == 0) multiply their accumulators by 2^-VD_F16X3_WSHIFT in the
 * epilogue (exact).  Valid for |w| < 2^(16 - VD_F16X3_WSHIFT) = 256. */
#define VD_F16X3_WSHIFT 8
#define VD_PREC_F16C8  4   /* fp16 operands + the two hi/lo CORRECTION products on the block-scaled fp8 matrix instruction
                              (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3, 2.2x the fp16 rate): a_hi W_hi in fp16, a_lo W_hi + a_hi W_lo in
                              fp8 once per four K steps -- two MFMA-equivalents per product instead of three, the corrections
                              exact to 2^-4 of themselves (2^-16 of the product).  The real side's LAST level only (pooled
                              features, 4 M tiles x 1 N tile per wave, K steps a multiple of 4): plane 1 of its source holds, per
                              16-byte slot, the 8 low parts (x 2^9) and the fp8 image of the 8 high parts (/ 4) as e4m3 bytes
                              (VdConvParams.emit_lo = 2 of the producing program), plane 1 of its packed weights the fp8
                              fragments of vd_pack_weights_c8 (the W_lo image is read, the W_hi image converted in registers),
                              out_scale points at that call's scales (two E8M0 codes, then s and 1 / s as floats).  The kernel
                              reads its B operands through running pointers up to six K steps past a channel chunk: the packed
                              buffer must be followed by 24 KB of readable memory.  Programs in position tiles: VD_EPI_POS_FEAT */

#define VD_EPI_POOL_CL   0
#define VD_EPI_POOL_FEAT 1
#define VD_EPI_ROWS      2
#define VD_EPI_POS_FEAT  3   /* VD_PREC_F16C8 programs in POSITION TILES (plan.plan_forward_pos): a tile's 32 rows are (clip, frame)
                                pairs of ONE output position, the four M tiles of a wave the four positions of one 2 x 2 pool
                                window; fp32 features out, pooled ACROSS the tiles (out table: 16 entries, one per row pair).  One
                                box type and one packed B operand set (w_box_stride apart) per window.  Behind the 2 S tap offsets
                                of every box type of a VD_PREC_F16C8 program follow S / 4 TILE SKIP MASKS: bit i of word g = M tile i
                                takes no part in K steps 4 g .. 4 g + 3 (its taps lie outside the input grid) */

/* One tile program (see video_distillation_amd/plan.py).  All pointers are device
 * pointers; strides are in the units given. */
typedef struct VdConvParams {
    const void* src;              /* 16-bit source, plane 0 (slots are 16 bytes at dword offsets) */
    int64_t src_plane_stride4;    /* dwords between the hi and lo planes (x3 precisions)       */
    int64_t src_clip_stride4;     /* dwords per clip                                           */
    int64_t src_chunk_stride4;    /* dwords per 8-channel chunk                                */
    const void* wpk;              /* packed weights [CC][S][NT][64][8] 16-bit, plane 0        */
    int64_t w_plane_stride;       /* 16-bit elements between hi and lo planes                 */
    int64_t w_box_stride;         /* 16-bit elements between the B operands of consecutive boxes (0: shared) */
    const float* bias;            /* [n_out] or NULL                                          */
    void* dst;                    /* POOL_CL: 16-bit slots plane 0; otherwise fp32            */
    int64_t dst_plane_stride;     /* POOL_CL: slots between hi and lo planes                  */
    uint8_t* argmax;              /* pooled epilogues: arg-max byte per output, or NULL        */
    const int32_t* col_off;       /* ROWS: element offset of column n, or NULL (= n*n_stride)  */
    const float* out_scale;       /* ROWS and select epilogues: device scalar the accumulators are multiplied by, or NULL */
    const int32_t* type_desc;     /* [ntypes][16]                                             */
    const int32_t* tables;        /* a_off / out / tap tables                                 */
    const int32_t* boxes;         /* [nbox][8]: a_off / out / tap table offsets, out origin, type */
    const int32_t* gather;        /* [nbox][gather_stride]: LDS slot -> source slot | clip<<24 */
    int64_t gather_stride;        /* multiple of 64 (one LDS-DMA wave-instruction = 64 slots)   */
    const void* zero_slot;        /* 16 zero bytes in device memory (source of zero fill)      */
    int32_t nbox, nclips, ncl;
    int32_t CC, S, NT, MW, MTW;
    int32_t epi, pool_t, relu, n_out, n_stride;
    int64_t out_clip_stride;
    int32_t out_chunk_stride, out_t_stride;
    int32_t lds_plane_bytes;
    int32_t prec;
    int32_t dbg;                  /* ablation switches for profiling (0 in production)         */
    int32_t ntypes;               /* number of box types; 1 -> tab_ofs are used for every box     */
    int32_t tab_ofs[3];           /* a_off / out / tap table offsets of box type 0                */
    int32_t atomic;               /* ROWS epilogue: accumulate with fp32 atomics                  */
    int32_t select;               /* pooled epilogues: argmax is an INPUT, emit the selected row (0 if ReLU-dead): acc * out_scale[0] + bias.
                                     2 (POOL_CL): as fp32 in the channels-last slot order ([clip][C/8][t][h][w][8] floats) instead of
                                     16-bit pairs -- the caller measures its range (vd_absmax_scale) and splits it (vd_split_scaled) */
    int32_t src_split_cc;         /* >0: channel chunks >= this come from a second tensor ...     */
    int64_t src_split_off4;       /* ... that starts this many dwords after src (same strides)    */
    int32_t NTW;                  /* N tiles per wave (0/1: one; 2: an A fragment feeds two MFMAs; wave columns = NT / NTW) */
    const int64_t* clip_index;    /* first-layer programs (ncl = 1): source clip of batch clip b is src + clip_index[b]*clip stride (NULL: b) */
    int32_t mt_valid;             /* NTW = 2: M tiles per box that carry rows (< MW*MTW: the last wave row skips its padding tile); 0 = all */
    int32_t persist;              /* 0: one workgroup per box; g>0: each workgroup walks boxes so that the grid is g generations of resident workgroups;
                                     vd_conv0_breg: bit 19 = run a frame-tile program (pair_flip != 0) with the plain K loop (one LDS read per
                                     MFMA) instead of the frame-sharing one (A/B measurements; bitwise the same results);
                                     bit 20 (hi+lo formats) = do NOT alternate the accumulation's sign per channel chunk: the matrix instruction rounds
                                     with a bias toward minus infinity (1.4 ulp per 600 v_mfma_f32_32x32x16_f16, whatever the sign), coherent over
                                     all outputs of a launch; by default a hi+lo program negates its accumulators at every chunk boundary and flips
                                     the sign of odd chunks' B fragments, which makes the bias zero-mean (sums over 1e8 outputs: 1e-5 -> 1e-6) */
    uint64_t* stamps;             /* dbg bit 3: [grid][8] s_memtime stamps of workgroup phases  */
    int32_t w_set_clips;          /* single-pass forward programs: > 0 = the B operand holds several sets, w_plane_stride elements apart; the box's
                                     first clip / w_set_clips picks the set (dithered real-side weights); 0 = one set */
    int32_t replica_stride;       /* atomic ROWS epilogue: dst += boxes[box][5] * replica_stride floats (weight-gradient programs spread their
                                     boxes over copies of dW so that same-address atomics do not serialise); 0 = one target */
    int32_t emit_lo;              /* single-pass programs with the staged POOL_CL epilogue (argmax NULL): != 0 also writes the LOW plane
                                     rn16(v - rn16(v)) of every pooled output, dst_plane_stride slots behind the high one -- the next level
                                     can then run in the hi+lo format of the same 16-bit type (real side: level 2 in f16x3); 2 = write plane 1
                                     for a VD_PREC_F16C8 consumer instead: per slot 8 e4m3 bytes of the low parts (x 2^9) and 8 of the high
                                     parts (/ 4), outputs clamped to +-1792 so that the image stays finite */
    int32_t src_planes, src_rows; /* first-level programs over pixel rows (vd_pix2rows): planes (frames x 3) and rows per plane of a clip; the
                                     kernel that builds its patch from aligned loads (vd_conv0_breg) derives row addresses
                                     and the zero fill from them instead of a gather table; 0 elsewhere */
    int32_t pair_flip;            /* pooled epilogues with pool_t = 1 (two outputs per row group, out_t_stride apart): bit q set = in the q-th
                                     row group of every MFMA tile the SECOND output lies out_t_stride BEFORE the first one (the out table
                                     holds the first).  0 everywhere except the first level's FRAME-TILE programs (plan.plan_forward_pix:
                                     0b0110 -- what makes their A-fragment reads free of LDS bank conflicts); != 0 also tells vd_conv0_breg
                                     that the four tiles of a wave row are the same positions in consecutive frames; bits 8..15 / 16..31 then
                                     carry the LDS pitches of the patch (slots per patch row / per plane), which that kernel's K loop has
                                     as instruction offsets */
    uint32_t* range_stats;        /* emit_lo = 2 producers: NULL, or two device words the launch updates -- [0] += pooled outputs that hit the
                                     +-1792 clamp (SATURATED: the consumer's operand is wrong there), [1] = max over launches of the
                                     float bits of max |output| (reset by the caller).  The fixed scalings of the fp8 operand planes keep
                                     the corrections of a VD_PREC_F16C8 consumer exact to 2^-16 for outputs of magnitude 2^-4 .. 1792
                                     (low parts below 2^-18 flush to zero: below 2^-7 an output's correction is gone, i.e. single-pass
                                     accuracy for it); a launch whose max |output| is below 0.25 or that saturates is outside the range
                                     the format was validated for -- distill.HipBackend.check_real_range then falls back to fp16 hi+lo */
} VdConvParams;

int vd_abi_version(void);

/* Conv3d(k(3,7,7), s(1,2,2), p(1,3,3)) [+bias +ReLU +MaxPool3d] forward, or its input
 * gradient, as one tile program.  Replaces nn.Conv3d / nn.ReLU / nn.MaxPool3d inside
 * ConvNet3D.features (networks.py:757, 768-770, 799) and their autograd backward w.r.t. the
 * input (distill_baseline.py:354, parameters frozen :336-337). */
int vd_conv_mfma(const VdConvParams* params, void* stream);
/* n <= 4 tile programs of the same shape class (prec, MTW, NT, MW; one N tile per wave; all plain or all accumulating / second-order)
 * in ONE launch -- the four parity classes of an input-gradient pass (the backward of the same nn.Conv3d, distill_baseline.py:354),
 * each of which alone starts too few workgroups to fill the chip at small batches.  Bitwise the results of n vd_conv_mfma calls.
 * -2: the programs do not share an instantiation (the caller then launches them one by one). */
int vd_conv_mfma_multi(const VdConvParams* const* params, int n, void* stream);

/* First-layer forward (x1 formats, one box type, NT=2 MW=2 MTW=4 S=32, pooled channels-last output, no arg-max, pixel-row source
 * with src_planes / src_rows set) with the layer's B fragments resident in registers across the workgroup's box walk: one
 * eight-wave workgroup per CU whose two groups of four waves alternate between the K loop of one box and everything else of
 * the next (row loads, kw-slot expansion, pool, output slots).  On frame-tile programs (pair_flip != 0) every A fragment is read
 * from LDS once for all the tiles of the wave it serves (68 reads per 128 MFMAs).  Same results as vd_conv_mfma on the same
 * program, bitwise (networks.py:799, the first nn.Conv3d + ReLU + MaxPool3d of ConvNet3D.features). */
int vd_conv0_breg(const VdConvParams* p, void* stream);

/* fp32 weights -> MFMA-fragment-ordered 16-bit operands (hi plane, and lo plane for the x3
 * precisions) through the planner's gather table.  n = number of packed elements. */
int vd_pack_weights(const float* w, const int32_t* widx, int64_t n, void* out_hi, void* out_lo,
                    int prec, void* stream);

/* Several vd_pack_weights calls as ONE launch (a training / trajectory-matching step packs the same weights for a dozen tile
   programs, each a 5 - 10 us launch): segment k is what vd_pack_weights(seg[k].w, seg[k].widx, seg[k].n, seg[k].out_hi, seg[k].out_lo,
   seg[k].prec) would do; bitwise the same outputs.  nseg <= VD_PACK_MAX. */
#define VD_PACK_MAX 24
typedef struct {
    const float* w; const int32_t* widx; int64_t n; void* out_hi; void* out_lo; int32_t prec; int32_t first_block;
} VdPackSeg;
typedef struct { int32_t nseg; int32_t reserved; VdPackSeg seg[VD_PACK_MAX]; } VdPackBatch;
int vd_pack_weights_multi(const VdPackBatch* batch, void* stream);
/* Operand planes of a VD_PREC_F16C8 program: out_hi = the fp16 fragments (as vd_pack_weights with VD_PREC_F16), out_c8 = the
 * same number of bytes holding, per four K steps, the fp8 (e4m3) fragments of W_hi x s and (W - W_hi) x s x 2^11 in the order the
 * kernel consumes them; s = the power of two that brings max|w| into [128, 256).  widx = the program's weight gather table
 * [CC][S][NT][64][8] (S a multiple of 4).  scales6 (6 floats of device scratch, caller-owned): words 0 and 1 receive, as int32, the
 * E8M0 scale codes VdConvParams.out_scale must point at when the program runs. */
int vd_pack_weights_c8(const float* w, int64_t w_elems, const int32_t* widx, int CC, int S, int NT, void* out_hi, void* out_c8,
                       float* scales6, void* stream);
/* Dithered single-pass weights for the real side of DM (f16 / bf16): out[g][i] = packed element i for launch group g of
 * `groups` (a power of two <= 64); every weight is rounded up in round(lam * groups) of the groups and down in the others, so
 * the mean over the groups equals the fp32 weight to 1/(2 groups) ulp and the weight-rounding perturbation of a class's mean
 * feature cancels to first order (the real clips of a class are dealt to the groups). */
int vd_pack_weights_dither(const float* w, const int32_t* widx, int64_t n, int groups, void* out, int prec, void* stream);

/* out[i] = (float) rn16(w[i]) in the 16-bit format of `prec` (f16 / bf16): the weights of the fresh
 * network of a DM iteration (get_network, distill_baseline.py:334) rounded once, so that the real-clip
 * forward (single-pass operands), the synthetic-clip forward (hi+lo operands) and the input gradient all
 * use identical weights.  out may alias w. */
int vd_round_operand(const float* w, int64_t n, int prec, float* out, void* stream);

/* (B,T,3,H,W) fp32 clips (the reference's input layout, networks.py:748) -> first-layer
 * source: 16-bit pixel rows [B][T*3][H][pitch], pitch = roundup8(W+8), row = 3 zeros, the W
 * pixels, zeros.  clip_index (optional, [nclips]) gathers batch clip b from x[clip_index[b]]:
 * the on-device form of get_images() (distill_baseline.py:84-90) over a pool resident in HBM. */
int vd_pix2rows(const float* x, const int64_t* clip_index, int64_t nclips, int T, int H, int W,
                void* out_hi, void* out_lo, int prec, void* stream);

/* Backward of ReLU + MaxPool3d: scatter the pooled gradient to the arg-max position of the
 * dense conv grid and emit it as channels-last slots (source of the input-gradient pass).
 * g_layout 0: g/argmax indexed [clip][n][pos] (embed features); 1: g [clip][pos][n],
 * argmax as channels-last bytes [clip][n/8][pos][8].
 * scale (optional device scalar): the gradient is multiplied by scale[0] before the 16-bit split. */
int vd_unpool_relu_bwd(const float* g, const uint8_t* argmax, int64_t nclips, int C, int To, int Ho, int Wo,
                       int pool_t, int T, int OH, int OW, int g_layout, void* out_hi, void* out_lo,
                       int prec, const float* scale, void* stream);

/* out[0] = 2^k with max|g| * 2^k in [target/2, target), out[1] = 2^-k (out[2] is scratch; out has
 * 4 floats).  Used to keep single-pass fp16 gradient operands inside fp16's exponent range. */
int vd_absmax_scale(const float* g, int64_t n, float target, float* out, void* stream);

/* DM class term, forward and gradient (distill_baseline.py:351):
 * loss[c] = sum_d (mean_b real[c,b,d] - mean_b syn[c,b,d])^2 ; g_syn = d loss / d syn. */
int vd_dm_loss(const float* feat_real, const float* feat_syn, int nclass, int nreal, int nsyn, int dim,
               float* loss_per_class, float* g_syn, void* stream);

/* out[g][d] = scale * sum_b x[g*per+b][d]: per-class partial sums of a rank's slice of the real
 * batch (torch.mean(output_real, dim=0), distill_baseline.py:351, split over ranks; the partial
 * sums are exchanged by one RCCL all-reduce). */
int vd_group_sum(const float* x, int groups, int per, int dim, float scale, float* out, void* stream);

/* torch.optim.SGD(momentum, dampening 0) step on the synthetic pixels
 * (distill_baseline.py:107, 355): buf = first ? g : mu*buf + g ; x -= lr*buf. */
int vd_sgd_momentum(float* x, float* buf, const float* g, int64_t n, float lr, float momentum, int first,
                    void* stream);

/* Conv3DNet 'concat' forward (utils.py:1186-1197) with the gathers of distill_s2d_ms.py:409-410
 * folded in: out[i] = conv3d(cat(static[sidx[i]] repeated over T, dynamic[didx[i]]), w, b).
 * static [ns][3][H][W], dynamic [nd][T][1][H][W], out [n][T][3][H][W]. sidx/didx may be NULL. */
int vd_hallucinator_fwd(const float* stat, const float* dyn, const int64_t* sidx, const int64_t* didx,
                        const float* w, const float* b, int n, int T, int H, int W, float* out, void* stream);

/* Its backward: g_dyn (scatter-added into [nd][T][1][H][W], pre-zeroed by the caller),
 * g_stat (optional, [ns][3][H][W], pre-zeroed), g_w [3*4*27], g_b [3] (pre-zeroed). */
int vd_hallucinator_bwd(const float* g_out, const float* stat, const float* dyn, const int64_t* sidx,
                        const int64_t* didx, const float* w, int n, int T, int H, int W,
                        float* g_dyn, float* g_stat, float* g_w, float* g_b, void* stream);

/* match_loss / distance_wb row reductions (utils.py:634-687).  For one gradient tensor viewed
 * as [rows][len]: acc[0] += sum_rows (1 - <r,s>/(|r||s|+1e-6)), acc[1] += sum (s-r)^2,
 * acc[2] += <r,s>, acc[3] += |r|^2, acc[4] += |s|^2.  acc is 5 fp32 (fp32 atomics). */
int vd_match_rows_fwd(const float* gr, const float* gs, int64_t rows, int len, float* acc, void* stream);
/* d/d gs of the three metrics; mode 0 'ours' (row cosine), 1 'mse', 2 'cos' (needs the global
 * sums in acc as produced by the forward).  gout = upstream scalar gradient. */
int vd_match_rows_bwd(const float* gr, const float* gs, int64_t rows, int len, int mode, const float* acc,
                      const float* gout, float* g_gs, void* stream);

/* The same for a whole list of gradient tensors (match_loss iterates the 8 parameter tensors of the network,
 * utils.py:660-664) in ONE launch: segment i = tensor pair i viewed as [rows][len]; `reserved` != 0 marks a FLAT segment
 * (rows = element count, len = 1: only the global sums of 'mse' / 'cos' are wanted; a row view of 'ours' with len == 1 --
 * a trailing unit axis, e.g. the (K,128,1,1,1) logit weights -- is NOT flat: each element is its own cosine row);
 * the backward writes d/d gs of segment i to seg[i].g.  VdMatchBatch.reserved: 0 = the forward adds all five sums; otherwise a
 * mask (bit k = acc[k] is wanted) -- the sums leave the launch as same-address atomics, one per block and sum, and a caller of
 * 'ours' (bit 0) or 'mse' (bit 1) needs one of the five. */
#define VD_MATCH_MAX_SEG 16
typedef struct VdMatchSeg {
    const float* gr;
    const float* gs;
    float* g;                     /* backward only */
    int64_t rows;
    int32_t len;
    int32_t reserved;
} VdMatchSeg;
typedef struct VdMatchBatch {
    int32_t nseg;
    int32_t reserved;
    VdMatchSeg seg[VD_MATCH_MAX_SEG];
} VdMatchBatch;
int vd_match_rows_fwd_multi(const VdMatchBatch* batch, float* acc, void* stream);
int vd_match_rows_bwd_multi(const VdMatchBatch* batch, int mode, const float* acc, const float* gout, void* stream);

/* Operand preparation for the weight-gradient tile program (plan.plan_wgrad):
 *  vd_clip_minor_cl : channels-last slots [plane][clip][C/8][npos][8 ch] -> clip-minor slots
 *                     [plane][c][ceil(nclips/8)][npos][8 clips] (zero for clips >= nclips);
 *  vd_clip_minor_pix: fp32 clips (B,T,3,H,W) -> [plane][c][ceil(B/8)][T][H][W][8 clips] 16-bit (hi, lo);
 *  vd_pack_dy       : dense dy slots [plane][clip][N/8][T][OH][OW][8] -> per-box MFMA B fragments
 *                     [plane][box][ceil(nclips/8)][S][N/32][64][8], box = block (nt,noh,now) of positions.
 * planes = 1 or 2 (hi, lo); strides in 16-byte slots. */
int vd_clip_minor_cl(const void* src, int64_t src_plane_slots, int planes, int64_t nclips, int C, int64_t npos,
                     void* dst, int64_t dst_plane_slots, void* stream);
int vd_clip_minor_pix(const float* x, int64_t nclips, int T, int H, int W, void* dst_hi, void* dst_lo, int prec,
                      void* stream);
int vd_pack_dy(const void* dy, int64_t dy_plane_slots, int planes, int64_t nclips, int N, int T, int OH, int OW,
               int nt, int noh, int now, void* dst, int64_t dst_plane_elems, void* stream);
/* vd_unpool_relu_bwd followed by vd_pack_dy, fused for a layer whose dense dy has no other reader (the first layer when no
 * pixel gradient is wanted): the packed B operand of the weight-gradient program straight from the pooled gradient and the
 * arg-max bytes, bitwise what the two calls produce.  (nt, noh, now) = the program's block of positions. */
int vd_unpool_relu_bwd_packed(const float* g, const uint8_t* argmax, int64_t nclips, int C, int To, int Ho, int Wo,
                              int pool_t, int T, int OH, int OW, int g_layout, int nt, int noh, int now,
                              void* dst_hi, void* dst_lo, int prec, const float* scale, void* stream);

/* Conv3d bias gradient from the POOLED gradient and the arg-max bytes of the layer (layouts as in vd_unpool_relu_bwd):
 * db[n] += sum of g over clips and pooled positions whose window was alive (bit 7 clear); accumulates (fp32 atomics). */
int vd_bias_grad_pooled(const float* g, const uint8_t* argmax, int64_t nclips, int C, int64_t npos, int g_layout, float* db,
                        void* stream);
/* The same with a FIXED summation order (bitwise reproducible): partial sums per (clip, block of 256 positions) into `scratch`
 * (vd_bias_grad_pooled_scratch_floats(nclips, C, npos) floats, caller-owned), folded in index order, added to db. */
int64_t vd_bias_grad_pooled_scratch_floats(int64_t nclips, int C, int64_t npos);
int vd_bias_grad_pooled_ordered(const float* g, const uint8_t* argmax, int64_t nclips, int C, int64_t npos, int g_layout,
                                float* scratch, float* db, void* stream);
/* db[n] += sum over clips and positions of dy (hi + lo planes) -- Conv3d bias gradient. */
int vd_bias_grad(const void* dy, int64_t dy_plane_slots, int planes, int64_t nclips, int N, int64_t npos, int prec,
                 const float* scale_inv, float* db, void* stream);

/* Classifier head of ConvNet3D.forward in inference (networks.py:738-745; evaluate_synset's test
 * passes, utils.py:793-824): AvgPool3d((kt,kh,kw), stride 1) over features (B,C,To,Ho,Wo), dropout
 * off, 1x1x1 conv (w [K][C], b [K]), squeeze, max over T -> logits (B,K). */
int vd_head_fwd(const float* feats, const float* w, const float* b, int64_t nclips, int C, int To, int Ho, int Wo,
                int kt, int kh, int kw, int K, float* out, void* stream);

/* Training half of evaluate_synset / epoch('train') (utils.py:765-792, 852-853):
 *  vd_standardize     : out = (x - mean(x)) / std(x), batch-global scalars, unbiased std (:770); scratch2 = 2 doubles;
 *  vd_head_train_fwd  : avg-pool, dropout (mask (B,C,Tp) holding 0 or 1/(1-p), or NULL), 1x1x1 conv, max over T;
 *                       also returns the dropped tensor (B,Tp,C) and the arg-max frame per (clip, class);
 *  vd_ce_loss         : nn.CrossEntropyLoss (mean): per-clip losses and d(mean loss)/d logits;
 *  vd_head_train_bwd  : gradients of the head: g_w [K][C], g_b [K] (accumulated, fp32 atomics), g_feats (B,C,To,Ho,Wo);
 *  vd_sgd_momentum_wd : torch.optim.SGD(momentum, weight_decay) step. */
int vd_standardize(const float* x, int64_t n, double* scratch2, float* out, void* stream);
int vd_head_train_fwd(const float* feats, const float* mask, const float* w, const float* b, int64_t nclips, int C, int To,
                      int Ho, int Wo, int kt, int kh, int kw, int K, float* dropped, float* logits, int32_t* amax_t, void* stream);
int vd_ce_loss(const float* logits, const int64_t* labels, int B, int K, float* loss_per_clip, float* dlogits, void* stream);
int vd_head_train_bwd(const float* dlogits, const int32_t* amax_t, const float* dropped, const float* mask, const float* w,
                      int64_t nclips, int C, int To, int Ho, int Wo, int kt, int kh, int kw, int K, float* g_w, float* g_b,
                      float* g_feats, void* stream);
/* Fixed-summation-order forms of the two calls above that accumulate with atomics (the deterministic training step, DESIGN
 * 8b): vd_standardize_ordered -- per-block partial sums in `scratch` (4096 doubles, caller-owned), folded in one fixed order
 * by every block; vd_head_train_bwd_ordered -- g_w / g_b as a gather over the clips in index order (no atomics).  Bitwise
 * reproducible from run to run; results differ from the atomic forms only in the last bits (summation order). */
int vd_standardize_ordered(const float* x, int64_t n, double* scratch, float* out, void* stream);
int vd_head_train_bwd_ordered(const float* dlogits, const int32_t* amax_t, const float* dropped, const float* mask, const float* w,
                              int64_t nclips, int C, int To, int Ho, int Wo, int kt, int kh, int kw, int K, float* g_w, float* g_b,
                              float* g_feats, void* stream);
/* Process-wide accumulation-order switch (initial value: environment VD_DETERMINISTIC=1, else 0).  While it is on,
 * weight-gradient programs PLANNED by the library (vd_program_build_wgrad, vd_train_create) give every box of positions its
 * own accumulation copy -- an fp32 atomic add onto a zeroed word with one contributor is exact, vd_replica_sum then folds the
 * copies in index order -- and vd_train_create'd handles use the *_ordered helpers: vd_train_step becomes bitwise
 * reproducible (reference: utils.py:765-792 on a CPU is deterministic for a seed and thread count).  The mode is captured
 * when a program / handle is created.  vd_set_deterministic returns the previous value. */
int vd_set_deterministic(int on);
int vd_get_deterministic(void);
/* Re-split 16-bit operand elements between the hi/lo formats (f16 pairs <-> bf16 pairs); lo pointers optional. */
int vd_resplit_slots(const void* src_hi, const void* src_lo, int64_t n_elems, int src_prec, void* dst_hi, void* dst_lo,
                     int dst_prec, void* stream);
/* fp32 -> 16-bit operand elements in the same order: dst_hi[i] (and dst_lo[i] for the hi+lo formats) = split of src[i] * scale[0]
 * (scale: device scalar, NULL = 1).  Turns the fp32 output of a select = 2 program into the scaled source of the next level of
 * the second-order sweep (the tangents d/dtheta of torch.autograd.grad(..., create_graph=True), distill_baseline.py:250). */
int vd_split_scaled(const float* src, int64_t n_elems, const float* scale, void* dst_hi, void* dst_lo, int prec, void* stream);
/* Device scalars of the power-of-two operand scales (a, b, out: 4-float blocks as vd_absmax_scale writes them -- scale, 1 / scale,
 * the float bits of max|g|): out[0] = a[0] * b[0] (mode 0) or min(a[0], b[0]) (mode 1: the common scale of two tensors that share
 * an accumulator; a tensor whose max|g| word is 0 -- all zeros -- constrains nothing), out[1] = 1 / out[0]; b NULL = 1. */
int vd_scale_combine(const float* a, const float* b, int mode, float* out, void* stream);
/* sha256 (first 16 hex digits + NUL) of the kernel sources and this header the library was BUILT from (compiled in by
 * video_distillation_amd/hip.py:build); the binding refuses a library whose stamp differs from its checkout's sources. */
const char* vd_sources_hash(void);

/* Second-order pass through the head for gradient matching (DC: match_loss(gw_syn, gw_real).backward() with
 * gw_syn = autograd.grad(CE(net(x)), params, create_graph=True), upstream DC loop / distill_baseline.py:250):
 * adjoints of the head's parameter gradients (v_w, v_b) and of the feature gradient (gbar_feats) -> adjoint of
 * the features (abar_feats), through the CE Hessian and the saved arg-max frames / dropout mask.  wbar [K][C] /
 * bbar [K] (optional, accumulated with fp32 atomics): adjoint of the head's own parameters -- the head rows of
 * the Hessian-vector product MTT's unrolled inner loop needs (distill_baseline.py:250-252).
 * logits == NULL: no loss Hessian (logitbar = 0) -- for a caller that differentiates its own loss on top of
 * net(x) (the autograd path of ConvNet3D.forward); dlogbar_out (optional, [nclips][K]) receives the adjoint of
 * dlogits, which that caller chains through its loss. */
int vd_head_second_order(const float* logits, const float* dlogits, const int32_t* amax_t, const float* dropped,
                         const float* mask, const float* w, const float* v_w, const float* v_b, const float* gbar_feats,
                         int64_t nclips, int C, int To, int Ho, int Wo, int kt, int kh, int kw, int K, float* abar_feats,
                         float* wbar, float* bbar, float* dlogbar_out, void* stream);
/* ---- planner in C++ (csrc/planner.cpp): no Python / offline step on the way to a launch ------------------------------
 * vd_program_build plans the FORWARD tile program of ConvNet3D level `layer` (0: Conv3d(3->64)+ReLU+MaxPool(1,2,2) over
 * pixel rows; 1, 2: Conv3d(->128)+ReLU+MaxPool(2,2,2) over channels-last slots; networks.py:792-814) for clips of
 * frames x height x width and operand precision `prec`, choosing box shape, LDS pitches, wave layout and -- for
 * batch_hint > 0 clips per launch -- the latency-oriented decomposition exactly as the Python planner does, and
 * returns the serialised program (malloc'd; vd_blob_free) that vd_program_load consumes. */
int vd_program_build(int layer, int frames, int height, int width, int prec, int batch_hint, void** blob, int64_t* nbytes);
/* the INPUT-GRADIENT programs of a level: level 0 -> one program (parity_class 0) that merges the four stride-2 parity
 * classes into the N dimension and writes fp32 pixels (T,3,H,W); levels 1, 2 -> one program per parity class
 * (parity_class = ph * 2 + pw) writing fp32 [t][h][w][cin].  Source of all: dense dy slots (vd_unpool_relu_bwd). */
int vd_program_build_dgrad(int layer, int parity_class, int frames, int height, int width, int batch_hint, void** blob,
                           int64_t* nbytes);
/* The weight-gradient program of ConvNet3D level `layer` for batches of `nclips` clips whose operands have `planes` 16-bit
 * planes (1: f16 / bf16, 2: the hi+lo formats) -- plan.plan_wgrad in C++ (byte-identical blob, tests/test_cplanner.py).
 * block3 receives the block of positions (nt, noh, now) vd_pack_dy / vd_unpool_relu_bwd_packed must pack dy for, *replicas
 * the number of accumulation copies ([copy][cin*147][cout] fp32, VdConvParams.replica_stride) vd_replica_sum folds into dW. */
int vd_program_build_wgrad(int layer, int frames, int height, int width, int nclips, int planes, void** blob, int64_t* nbytes,
                           int* block3, int* replicas);
void vd_blob_free(void* blob);

/* ---- ConvNet3D.embed (networks.py:747-751) as one handle: nothing but this header, the library and device pointers.
 * The handle owns the three planned programs and their packed weights; clips / features / workspace are caller-owned.
 *   vd_embed_create          plans + loads the three forward programs of the geometry (batch_hint: typical clips per call, 0 = large)
 *   vd_embed_workspace_bytes bytes of scratch vd_embed_forward needs for nclips clips (pixel rows + two activation tensors)
 *   vd_embed_set_weights     packs w0, w1, w2 (fp32 device tensors, the reference's Conv3d layouts) into MFMA operands; b0..b2
 *                            (fp32 device vectors) are read by the launches and must stay alive
 *   vd_embed_forward         features[nclips][vd_embed_num_features] = embed(clips[(clip_index ? clip_index[b] : b)]),
 *                            clips (.,T,3,H,W) fp32 -- the reference's input layout; asynchronous on `stream`
 * Errors: -1 bad argument, -6 no weights set, -7 workspace too small, otherwise the failing call's code.
 * With the INPUT GRADIENT (the DM class term's backward to the synthetic pixels, distill_baseline.py:351-354):
 *   vd_embed_create_ex       also plans + loads the input-gradient programs with operand precision prec_bwd (-1: none)
 *   vd_embed_forward_keep    forward that also stores the pooling arg-max bytes of the three levels in `argmax`
 *                            (vd_embed_argmax_bytes(e, nclips) bytes, caller-owned)
 *   vd_embed_backward        g_clips (nclips,T,3,H,W) fp32 = d <g_features, features> / d clips for such a forward
 *                            (vd_embed_backward_workspace_bytes of scratch); error -8 if created without prec_bwd */
typedef struct VdEmbed VdEmbed;
int vd_embed_create(int frames, int height, int width, int prec, int batch_hint, VdEmbed** out);
int vd_embed_create_ex(int frames, int height, int width, int prec, int prec_bwd, int batch_hint, VdEmbed** out);
int64_t vd_embed_argmax_bytes(const VdEmbed* e, int64_t nclips);
int64_t vd_embed_backward_workspace_bytes(const VdEmbed* e, int64_t nclips);
int vd_embed_forward_keep(VdEmbed* e, const float* clips, const int64_t* clip_index, int64_t nclips, void* workspace,
                          int64_t workspace_bytes, float* features, uint8_t* argmax, void* stream);
int vd_embed_backward(VdEmbed* e, const float* g_features, const uint8_t* argmax, int64_t nclips, void* workspace,
                      int64_t workspace_bytes, float* g_clips, void* stream);
int64_t vd_embed_num_features(const VdEmbed* e);
int64_t vd_embed_workspace_bytes(const VdEmbed* e, int64_t nclips);
int vd_embed_set_weights(VdEmbed* e, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                         const float* b2, void* stream);
int vd_embed_forward(VdEmbed* e, const float* clips, const int64_t* clip_index, int64_t nclips, void* workspace,
                     int64_t workspace_bytes, float* features, void* stream);
void vd_embed_free(VdEmbed* e);

/* MFMA-saturating microbenchmark: blocks x 4 waves each issue iters x 8 groups of 32768 FLOP from registers on
 * pseudo-random operands (shape 0: v_mfma_f32_32x32x16_f16; 1: pairs of v_mfma_f32_16x16x32_f16); out receives
 * blocks*256 floats.  The measured dense 16-bit peak that bench.py prints beside the 2.5 PFLOP/s spec figure
 * (BASELINE.md section 3). */
int vd_mfma_peak(int blocks, int iters, int shape, float* out, void* stream);
int vd_sgd_momentum_wd(float* x, float* buf, const float* g, int64_t n, float lr, float momentum, float wd, int first,
                       void* stream);

/* Folds the accumulation copies of a weight-gradient tile program back into the gradient tensor:
 * out[col][row] += sum over r of rep[r][row][col], rep = (replicas, rows = cin*147, cols = cout) fp32, out = dW (cout, cin, 3, 7, 7).
 * The programs accumulate cout-minor (the 32 lanes of an atomic instruction share one 128-byte line) and spread their
 * boxes over the copies (VdConvParams.replica_stride = rows*cols; box row word 5 = the copy).  The copies are SCRATCH: more
 * than 64 of them (the ordered mode's one copy per box) are first folded in place, 32 consecutive copies into the first of
 * each group, then the group heads -- always in index order, no atomics: the result is bitwise reproducible. */
int vd_replica_sum(float* rep, int replicas, int rows, int cols, float* out, void* stream);

/* Decoded frames -> clips, the device half of the dataset preload (replaces the per-frame host transform
 * `ToTensor()` + `Normalize(mean, std)` of utils.py:171-173 and the per-step host->device copy of get_images,
 * distill_baseline.py:84-90): src (nframes, H, W, 3) uint8 -> dst (nframes, 3, H, W) fp32,
 * dst = (src / 255 - mean[c]) / std[c], each operation correctly rounded (bit-equal to the host transform).
 * mean3 / std3 are HOST pointers to 3 floats; std must be non-zero.  HBM-bound: 15 bytes per pixel. */
int vd_frames_normalize(const void* src_u8, float* dst, int64_t nframes, int height, int width, const float* mean3,
                        const float* std3, void* stream);

/* ---- Serialised tile programs: the torch-free, Python-free way to run a layer ------------------------------
 * A program blob is written offline by the planner (engine.export_program(plan) / tools/export_programs.py) for
 * one layer geometry; it replaces what nn.Conv3d/nn.ReLU/nn.MaxPool3d's constructors hold in the reference
 * (networks.py:792-814) plus the launch geometry.  The handle owns only the program's small device tables and the
 * packed-weight buffer; sources, weights, bias, outputs and arg-max are caller-owned device memory.
 *  vd_program_load         : parse + upload (prec = VD_PREC_*); returns 0 and *out, or <0 / a hipError_t;
 *  vd_program_pack_weights : fp32 Conv3d weight (device) -> this program's MFMA B fragments;
 *  vd_program_run          : launch for nclips clips (src_plane_slots / dst_plane_stride as in VdConvParams;
 *                            clip_index optional, first-layer programs only);
 *  vd_program_info         : 0 boxes per clip group, 1 packed weight elements per plane, 2 planes,
 *                            3 output clip stride, 4 output channels;
 *  vd_program_free         : release the handle. */
typedef struct VdProgram VdProgram;
int vd_program_load(const void* blob, int64_t nbytes, int prec, VdProgram** out);
int vd_program_pack_weights(VdProgram* prog, const float* w, void* stream);
int vd_program_run(VdProgram* prog, const void* src, int64_t src_plane_slots, const float* bias, void* dst,
                   int64_t dst_plane_stride, uint8_t* argmax, const int64_t* clip_index, int nclips, void* stream);
/* the same with the fp32 epilogue scale of the input-gradient programs (out_scale[0] multiplies every output; NULL = 1) */
int vd_program_run_scaled(VdProgram* prog, const void* src, int64_t src_plane_slots, const float* bias, void* dst,
                          int64_t dst_plane_stride, uint8_t* argmax, const int64_t* clip_index, int nclips,
                          const float* out_scale, void* stream);
/* Run a weight-gradient program (vd_program_build_wgrad + vd_program_load): source = x in the clip-minor layout of
 * vd_clip_minor_cl / _pix, B operand = the packed dy of vd_pack_dy / vd_unpool_relu_bwd_packed for the program's block,
 * `copies` = [replicas][cin*147][cout] fp32 zeroed by the caller (copy_elems = cin*147*cout), accumulated into with fp32
 * atomics and folded into dW by vd_replica_sum.  out_scale: device float multiplied into the result (NULL = 1). */
int vd_program_run_wgrad(VdProgram* prog, const void* x_clip_minor, int64_t x_plane_slots, const void* packed_dy,
                         int64_t packed_plane_elems, float* copies, int64_t copy_elems, int cin, const float* out_scale,
                         void* stream);
int64_t vd_program_info(const VdProgram* prog, int what);
void vd_program_free(VdProgram* prog);

/* ---- One evaluate_synset training step (utils.py:765-792, 852-853) behind one handle, no Python ---------------------------
 * forward with kept activations, classifier head (AvgPool3d, dropout mask, 1x1x1 conv, max over frames), CrossEntropyLoss,
 * backward (bias / weight / input gradients per level) and torch.optim.SGD(momentum, weight_decay) on the 8 parameter tensors.
 * prec / prec_bwd: operand formats of the forward and the gradient passes (same 16-bit family; the gradient passes read the
 * forward's kept activations, so prec_bwd may not have more planes than prec).  nclips is fixed at creation (the programs are
 * planned for it).  The caller owns all tensors and the workspace (vd_train_workspace_bytes).
 * vd_train_step: params[8] / momentum[8] device fp32 in parameters() order, updated in place; `first` != 0 initialises the
 * momentum buffers with the gradient (SGD's first step); clips (nclips, T, 3, H, W) already standardised (vd_standardize);
 * labels int64; dropout_mask (nclips, 128, T') of 0 or 1/(1-p), or NULL; loss_per_clip (nclips) / logits (nclips, K) optional.
 * Errors: -1 arguments, -2 precisions do not combine, -7 workspace too small, -9 runtime. */
typedef struct VdTrain VdTrain;
int vd_train_create(int frames, int height, int width, int num_classes, int prec, int prec_bwd, int64_t nclips, VdTrain** out);
int64_t vd_train_workspace_bytes(const VdTrain* t);
int vd_train_step(VdTrain* t, float* const* params, float* const* momentum, const float* clips, const int64_t* labels,
                  const float* dropout_mask, float lr, float momentum_coef, float weight_decay, int first, void* workspace,
                  int64_t workspace_bytes, float* loss_per_clip, float* logits, void* stream);
void vd_train_free(VdTrain* t);

/* ---- Exchange steps of the sharded loops (one process per GPU), over RCCL / xGMI -------------------------------------------
 * Replaces what nn.DataParallel does for the reference (utils.py:615-623: scatter the batch, replicate the module, gather the
 * outputs over PCIe through GPU 0) by explicit collectives on device buffers; the Python trainers issue the same ones through
 * torch.distributed (distill.py `_all_reduce` / `_all_gather`).  Per DM step the only data-path exchange is the all-reduce of
 * the per-class feature sums of split classes (hybrid / batch decompositions: 2048 floats per class; pre-scale them with
 * vd_group_sum's `scale`); s2d adds the 327 hallucinator gradients; MTT the flat parameter gradient and the Hessian-vector
 * product per student step; vd_comm_allgather_f32 brings the class-owned synthetic clips together before evaluate_synset.
 * librccl.so is resolved at the first call (dlopen): -10 = not available on this box, -11 = RCCL reported an error.
 * vd_comm_unique_id: rank 0 creates the id and passes its 128 bytes to the other ranks over any host channel; every rank then
 * calls vd_comm_create with ITS HIP device current.  Collectives are asynchronous on `stream`; send == recv is allowed. */
typedef struct VdComm VdComm;
typedef struct { char internal[128]; } VdCommId;
int vd_comm_unique_id(VdCommId* id);
int vd_comm_create(const VdCommId* id, int nranks, int rank, VdComm** out);
int vd_comm_size(const VdComm* c);                 /* ncclCommCount of the communicator */
int vd_comm_version(int* version);                 /* ncclGetVersion (major * 10000 + minor * 100 + patch) */
int vd_comm_rank(const VdComm* c);
int vd_comm_allreduce_f32(VdComm* c, const float* send, float* recv, int64_t n, void* stream);
int vd_comm_allgather_f32(VdComm* c, const float* send, float* recv, int64_t n_per_rank, void* stream);
void vd_comm_free(VdComm* c);

#ifdef __cplusplus
}
#endif
#endif /* VD_HIP_H */
