"""Replay the tile programs produced by video_distillation_amd.plan on the CPU (numpy fp64,
tests/emulate.py) and compare with the oracle.  This pins every table the HIP kernel consumes
(row origins, tap offsets, weight gather, output maps) without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from tests import emulate as E
from video_distillation_amd import plan as P

GEO = P.NetGeometry(8, 64, 64)


@pytest.fixture(scope="module")
def net():
    return P.plan_network(GEO)


@pytest.fixture(scope="module")
def acts():
    params = [p.double() for p in R.init_params(5)]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 8, 3, 64, 64, generator=g).double()
    col = []
    R.feature_layers(x.permute(0, 2, 1, 3, 4), params, collect=col)
    return params, x, col


def cl_to_bcthw(flat, nclips, shape):
    ccs, T, Ho, Wo, _ = shape
    a = flat.reshape(nclips, ccs, T, Ho, Wo, 8)
    return a.transpose(0, 1, 5, 2, 3, 4).reshape(nclips, ccs * 8, T, Ho, Wo)


def bcthw_to_cl(a):
    B, C, T, H, W = a.shape
    return a.reshape(B, C // 8, 8, T, H, W).transpose(0, 1, 3, 4, 5, 2).copy()


def test_plans_cover_and_fit(net):
    for pl in net["fwd"] + [p for l in net["bwd"] for p in l]:
        assert pl.lds_slots * 16 * 2 + 4096 <= 160 * 1024, pl.name  # hi+lo planes must fit LDS
        assert pl.MTW in (2, 3, 4, 7, 8) and pl.threads in (64, 128, 256)
        for t in pl.types:
            assert t.a_off.min() >= 0 and (t.a_off.max() + t.tap_off.max()) // 16 < pl.ncl * t.pitch_c + 1
            assert t.a_off.size == pl.MW * pl.MTW * 32


def test_forward_layer0(net, acts):
    params, x, col = acts
    pl = net["fwd"][0]
    n = 2
    src = E.pix_to_rows(x[:n].numpy())
    out = np.zeros(n * int(np.prod(pl.out_shape)))
    arg = E.run_plan(pl, src, params[0].numpy().ravel(), params[1].numpy(), n, out)
    got = cl_to_bcthw(out, n, pl.out_shape)
    np.testing.assert_allclose(got, col[2][:n].numpy(), rtol=1e-9, atol=1e-9)
    assert len(arg) == out.size


def test_forward_layer1(net, acts):
    params, x, col = acts
    pl = net["fwd"][1]
    n = 1
    src = bcthw_to_cl(col[2][:n].numpy())
    out = np.zeros(n * int(np.prod(pl.out_shape)))
    E.run_plan(pl, src, params[2].numpy().ravel(), params[3].numpy(), n, out)
    np.testing.assert_allclose(cl_to_bcthw(out, n, pl.out_shape), col[5][:n].numpy(), rtol=1e-9, atol=1e-9)


def test_forward_layer2_features_and_ragged_clip_group(net, acts):
    params, x, col = acts
    pl = net["fwd"][2]
    n = 3 if pl.ncl > 1 else 1           # 3 clips with ncl=4: last group is ragged
    src = bcthw_to_cl(col[5][:n].numpy())
    out = np.zeros(n * GEO.num_feat)
    arg = E.run_plan(pl, src, params[4].numpy().ravel(), params[5].numpy(), n, out)
    want = col[8][:n].numpy().reshape(n, -1)
    np.testing.assert_allclose(out.reshape(n, -1), want, rtol=1e-9, atol=1e-9)
    # arg-max index convention: j = dt*4 + dh*2 + dw inside the 2x2x2 window
    relu2 = col[7][0].numpy()            # (C, T, H, W) post-ReLU conv grid of clip 0
    C, T, H, W = relu2.shape
    win = relu2.reshape(C, T // 2, 2, H // 2, 2, W // 2, 2).transpose(0, 1, 3, 5, 2, 4, 6).reshape(C, -1, 8)
    am = win.argmax(axis=2).ravel()
    got = np.array([arg[i] for i in range(GEO.num_feat)])
    np.testing.assert_array_equal(got, am)


@pytest.mark.parametrize("n", [3, 9])
def test_forward_layer2_in_position_tiles(net, acts, n):
    """plan_forward_pos: rows = (clip, frame) pairs of one output position, taps skipped per tile, pool across the four tiles;
    ragged clip groups (n % ncl != 0).  Every tap the program multiplies reads inside the un-padded patch (asserted by the emulator)."""
    params, x, col = acts
    dims = net["dims"][2]
    pl = P.plan_forward_pos("fwd2_pos", dims[0], dims[1], dims[2], dims[3], dims[4], dims[11])
    assert pl.epi == P.EPI_POS_FEAT and pl.ncl * dims[5] == 32 and pl.w_box_stride == pl.CC * pl.S * pl.NT * 512
    assert all(t.tap_off.size == 2 * pl.S + pl.S // 4 for t in pl.types)
    g = torch.Generator().manual_seed(5)
    a1 = torch.relu(torch.randn(n, dims[0], dims[2], dims[3], dims[4], generator=g).double())
    y = F.conv3d(a1, params[4], params[5], stride=(1, 2, 2), padding=(1, 3, 3))
    want = F.max_pool3d(torch.relu(y), (2, 2, 2)).reshape(n, -1).numpy()
    out = np.zeros(n * GEO.num_feat)
    E.run_plan(pl, bcthw_to_cl(a1.numpy()), params[4].numpy().ravel(), params[5].numpy(), n, out)
    np.testing.assert_allclose(out.reshape(n, -1), want, rtol=1e-9, atol=1e-9)


def test_forward_layer2_in_position_tiles_at_the_benchmark_geometry():
    """The same at 112x112x16 (7 x 7 grid, 8 frames: four clips per box, masks 0 / 0b1010 / 0b1100 / 0b1110 only), 5 clips."""
    dims = P.NetGeometry(16, 112, 112).layer_dims()[2]
    pl = P.plan_forward_pos("fwd2_pos", dims[0], dims[1], dims[2], dims[3], dims[4], dims[11])
    masks = {int(m) for t in pl.types for m in t.tap_off[2 * pl.S:]}
    assert masks == {0, 0xA, 0xC, 0xE}
    g = torch.Generator().manual_seed(6)
    n = 5
    w = torch.randn(128, 128, 3, 7, 7, generator=g).double() * 0.02
    b = torch.randn(128, generator=g).double() * 0.1
    a1 = torch.relu(torch.randn(n, dims[0], dims[2], dims[3], dims[4], generator=g).double())
    want = F.max_pool3d(torch.relu(F.conv3d(a1, w, b, stride=(1, 2, 2), padding=(1, 3, 3))), (2, 2, 2)).reshape(n, -1).numpy()
    out = np.zeros(n * want.shape[1])
    E.run_plan(pl, bcthw_to_cl(a1.numpy()), w.numpy().ravel(), b.numpy(), n, out)
    np.testing.assert_allclose(out.reshape(n, -1), want, rtol=1e-9, atol=1e-9)


def test_position_tiles_at_the_benchmark_geometry_halve_the_matrix_work():
    dims = P.NetGeometry(16, 112, 112).layer_dims()[2]
    pl = P.plan_forward_pos("fwd2_pos", dims[0], dims[1], dims[2], dims[3], dims[4], dims[11])
    old = P.plan_forward_cl("fwd2", dims[0], dims[1], dims[2], dims[3], dims[4], dims[11], feat_out=True, mtw_options=(4,), step_multiple=4)
    assert (pl.S, pl.ncl, pl.nbox) == (56, 4, 4) and pl.meta["mfma_per_chunk"] == [152] * 4
    per_clip_new = sum(pl.meta["mfma_per_chunk"]) / pl.ncl
    per_clip_old = old.S * old.MW * old.MTW * old.nbox / old.ncl
    assert per_clip_new / per_clip_old == 0.5
    assert all(t.conflict_cycles == 4.0 for t in pl.types)


@pytest.mark.parametrize("li", [0, 1, 2])
def test_input_gradient_passes(net, acts, li):
    params, x, col = acts
    cin, cout, t, h, w, T, OH, OW = net["dims"][li][:8]
    g = torch.Generator().manual_seed(900 + li)
    n = 2 if li == 2 else 1
    dy = torch.randn(n, cout, T, OH, OW, generator=g).double()
    xin = torch.zeros(n, cin, t, h, w, dtype=torch.double, requires_grad=True)
    y = F.conv3d(xin, params[2 * li], None, stride=(1, 2, 2), padding=(1, 3, 3))
    (want,) = torch.autograd.grad(y, xin, dy)
    src = bcthw_to_cl(dy.numpy())
    out = np.zeros(n * cin * t * h * w)
    for pl in net["bwd"][li]:
        E.run_plan(pl, src, params[2 * li].numpy().ravel(), None, n, out)
    if li == 0:      # pixel layout (T, C, H, W)
        got = out.reshape(n, t, cin, h, w).transpose(0, 2, 1, 3, 4)
    else:            # (T, H, W, C)
        got = out.reshape(n, t, h, w, cin).transpose(0, 4, 1, 2, 3)
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-9, atol=1e-9)


def test_first_level_input_gradient_in_2x4_pixel_blocks():
    """plan_dgrad_pix(bw=4): one GEMM row = a 2 x 4 pixel block (N = 24 of 32 columns, 3 x 4 x 5 = 60 taps) instead of a 2 x 2
    block (N = 12, 48 taps) -- half the rows for 1.25x the K steps: 0.625 of the MFMA work at 112 x 112.  Both decompositions,
    full-size and small-box forms, against autograd of the reference's Conv3d (networks.py:757)."""
    import torch.nn.functional as F
    cin, cout, t, h, w = 3, 64, 4, 24, 32
    g = torch.Generator().manual_seed(11)
    wt = torch.randn(cout, cin, 3, 7, 7, generator=g).double()
    dy = torch.randn(1, cout, t, (h - 1) // 2 + 1, (w - 1) // 2 + 1, generator=g).double()
    xin = torch.zeros(1, cin, t, h, w, dtype=torch.double, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv3d(xin, wt, None, stride=(1, 2, 2), padding=(1, 3, 3)), xin, dy)
    for bw, n_out, taps in ((2, 12, 48), (4, 24, 60)):
        for kw in (dict(), dict(lds_budget=1800, mtw_options=(4,))):
            pl = P.plan_dgrad_pix("bwd0_merged", cin, cout, t, h, w, bw=bw, **kw)
            assert pl.n_out == n_out and pl.S == taps // 2 and pl.meta["block_w"] == bw
            out = np.zeros(cin * t * h * w)
            E.run_plan(pl, bcthw_to_cl(dy.numpy()), wt.numpy().ravel(), None, 1, out)
            np.testing.assert_allclose(out.reshape(1, t, cin, h, w).transpose(0, 2, 1, 3, 4), want.numpy(), rtol=1e-9, atol=1e-9)
    full2 = P.plan_dgrad_pix("bwd0_merged", 3, 64, 16, 112, 112, bw=2)
    full4 = P.plan_dgrad_pix("bwd0_merged", 3, 64, 16, 112, 112, bw=4)
    work = lambda pl: pl.rows_total * pl.S          # noqa: E731   MFMA tiles x K steps
    assert work(full4) < 0.7 * work(full2), (work(full4), work(full2))
    assert P.bwd0_block_w(112) == 4 and P.bwd0_block_w(110) == 2


def test_full_resolution_plans_build():
    net = P.plan_network(P.NetGeometry(16, 112, 112))
    f1 = net["fwd"][1]
    assert f1.rows_useful == 16 * 14 * 14 and f1.rows_total == f1.rows_useful  # exact tiling, no wasted MFMA rows
    assert net["fwd"][2].ncl == 2
    for pl in net["fwd"] + [p for l in net["bwd"] for p in l]:
        assert pl.lds_slots * 32 + 4096 <= 160 * 1024


def test_odd_geometry_floor_pooling_and_multi_type_boxes():
    """12 x 96 x 80 clips: conv grids of 12x12x10 and 6x3x3 (odd extents -> floor pooling drops a
    row/column, ragged edge boxes, several box types per plan, up to 7 clips per workgroup)."""
    import torch.nn.functional as F
    geo = P.NetGeometry(12, 96, 80)
    net = P.plan_network(geo)
    assert geo.num_feat == 384
    d0 = net["dims"][0]
    ragged = P.plan_dgrad_pix("bwd0_merged", d0[0], d0[1], d0[2], d0[3], d0[4])     # the 7 / 8-tile fallback decomposition of the
    assert len(ragged.types) > 1                                                    # pixel-gradient program: several box types
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 12, 3, 96, 80, generator=g).double()
    params = [p.double() for p in R.init_params(5)]
    col=[]; R.feature_layers(x.permute(0,2,1,3,4), params, collect=col)
    cl_to, to_cl = cl_to_bcthw, bcthw_to_cl
    n=2
    pl=net['fwd'][0]; out=np.zeros(n*int(np.prod(pl.out_shape))); E.run_plan(pl,E.pix_to_rows(x.numpy()),params[0].numpy().ravel(),params[1].numpy(),n,out)
    np.testing.assert_allclose(cl_to(out,n,pl.out_shape), col[2].numpy(), rtol=1e-9, atol=1e-9)
    pl=net['fwd'][1]; out=np.zeros(n*int(np.prod(pl.out_shape))); E.run_plan(pl,to_cl(col[2].numpy()),params[2].numpy().ravel(),params[3].numpy(),n,out)
    np.testing.assert_allclose(cl_to(out,n,pl.out_shape), col[5].numpy(), rtol=1e-9, atol=1e-9)
    pl=net['fwd'][2]; out=np.zeros(n*geo.num_feat); E.run_plan(pl,to_cl(col[5].numpy()),params[4].numpy().ravel(),params[5].numpy(),n,out)
    np.testing.assert_allclose(out.reshape(n,-1), col[8].numpy().reshape(n,-1), rtol=1e-9, atol=1e-9)
    dims=net['dims']
    for li in range(3):
        cin,cout,t_,h,w,T,OH,OW=dims[li][:8]
        dy=torch.randn(1,cout,T,OH,OW,generator=g).double()
        xin=torch.zeros(1,cin,t_,h,w,dtype=torch.double,requires_grad=True)
        y=F.conv3d(xin,params[2*li],None,stride=(1,2,2),padding=(1,3,3)); (want,)=torch.autograd.grad(y,xin,dy)
        out=np.zeros(cin*t_*h*w)
        for pl in net['bwd'][li]: E.run_plan(pl,to_cl(dy.numpy()),params[2*li].numpy().ravel(),None,1,out)
        got = out.reshape(1,t_,cin,h,w).transpose(0,2,1,3,4) if li==0 else out.reshape(1,t_,h,w,cin).transpose(0,4,1,2,3)
        np.testing.assert_allclose(got,want.numpy(),rtol=1e-9,atol=1e-9)
        if li == 0:      # ... the multi-type decomposition of the same program, and the small-box one of the training engines
            small = P.plan_network(geo, bwd0_small=True)["bwd"][0][0]
            assert small.MTW == 4 and small.nbox > ragged.nbox
            for alt in (ragged, small):
                out=np.zeros(cin*t_*h*w)
                E.run_plan(alt,to_cl(dy.numpy()),params[0].numpy().ravel(),None,1,out)
                np.testing.assert_allclose(out.reshape(1,t_,cin,h,w).transpose(0,2,1,3,4),want.numpy(),rtol=1e-9,atol=1e-9)


def test_exported_program_blob_layout():
    """plan.export_program: 40 int64 header words + the int32 arrays whose lengths the header lists
    (what vd_program_load in csrc/program.hip parses)."""
    from video_distillation_amd import plan as P
    net = P.plan_network(P.NetGeometry(8, 64, 64), ntw=2, ntw0=2, balanced=True)
    for pl in net["fwd"] + [net["bwd"][1][0]]:
        blob = P.export_program(pl)
        h = np.frombuffer(blob[:P.PROGRAM_HEADER_WORDS * 8], dtype=np.int64)
        assert blob[:8] == P.PROGRAM_MAGIC
        assert (h[1], h[2], h[3], h[4], h[5], h[6]) == (pl.CC, pl.S, pl.NT, pl.MW, pl.MTW, pl.NTW)
        n_int = int(h[28:34].sum())
        assert len(blob) == P.PROGRAM_HEADER_WORDS * 8 + 4 * n_int
        ints = np.frombuffer(blob[P.PROGRAM_HEADER_WORDS * 8:], dtype=np.int32)
        o = int(h[28] + h[29] + h[30])
        np.testing.assert_array_equal(ints[o:o + int(h[31])], pl.gather_table().reshape(-1))
        np.testing.assert_array_equal(ints[o + int(h[31]):o + int(h[31]) + int(h[32])], pl.widx.reshape(-1))
        assert h[22] == pl.ncl and h[23] == pl.nbox and h[24] == pl.gather_table().shape[1]


@pytest.mark.parametrize("geom", [(16, 112, 112), (8, 64, 64), (12, 96, 80), (6, 48, 64), (10, 40, 48)])
def test_first_level_frame_tiles_have_the_structure_the_sharing_kernel_relies_on(geom):
    """plan_forward_pix (round 5): wherever the pooled width is a multiple of 4 the first level is planned in FRAME TILES -- the four
    M tiles of a wave row are the same 32 positions in consecutive frames, K step 3 j + kt holds tap pair j of kernel plane kt with
    both taps in that plane, the A-fragment reads are free of LDS bank conflicts at pitches 9 / 189, and ``pair_flip`` carries the
    flip mask + the pitches -- which is what lets conv0_breg_kernel<PREC, true> read one fragment for up to three MFMAs with
    instruction-offset taps.  Geometries that do not split that way fall back to frame-pair row groups."""
    t_in, h, w = geom
    pl = P.plan_forward_pix("fwd0", 64, t_in, h, w)
    OW = P.conv_out_dim(w, P.KW, 2, 3)
    if (OW // 2) % 4 != 0 or pl.meta["box"] != (4, 8, 8):        # (e.g. 6 frames: the box chooser takes all six, 6 x 4 x 8)
        assert pl.pair_flip == 0 and pl.out_t_stride == (P.conv_out_dim(h, P.KH, 2, 3) // 2) * (OW // 2) and "frame_tiles" not in pl.meta
        assert geom in ((6, 48, 64), (10, 40, 48))
        return
    assert pl.meta.get("frame_tiles") == 1 and (pl.NT, pl.MW, pl.MTW, pl.S) == (2, 2, 4, 32)
    assert pl.pair_flip == P.FRAME_TILE_FLIP | (9 << 8) | (189 << 16) and pl.out_t_stride == P.FRAME_TILE_OUT_STEP
    for t in pl.types:
        assert (t.pitch_h, t.pitch_f) == (9, 189) and t.conflict_cycles == 4.0
        a = t.a_off.reshape(2, 4, 32)
        for i in range(4):                                       # tile i = tile 0 one frame (three planes) further
            assert (a[:, i] - a[:, 0] == i * 3 * 189 * 16).all()
        tp = t.tap_off.reshape(32, 2)
        for j in range(10):
            q0, q1 = 2 * j, 2 * j + 1                            # (c, kh) = divmod(q, 7): both taps of a pair in one kernel plane
            want = [((q // 7) * 189 + (q % 7) * 9) * 16 for q in (q0, q1)]
            for kt in range(3):
                assert tp[3 * j + kt].tolist() == [v + kt * 3 * 189 * 16 for v in want]
        last = (2 * 189 + 6 * 9) * 16                            # q = 20: (c, kh) = (2, 6)
        assert tp[30].tolist() == [last, last + 3 * 189 * 16] and tp[31].tolist() == [last + 2 * 3 * 189 * 16, 0]
    # every output position appears exactly once over the out tables of a clip (both windows of every row group)
    T = P.conv_out_dim(t_in, P.KT, 1, 1); Ho = P.conv_out_dim(h, P.KH, 2, 3) // 2; Wo = OW // 2
    seen = np.zeros(T * Ho * Wo, dtype=np.int64)
    for box in pl.boxes:
        t = pl.types[int(box[0])]
        for k, o in enumerate(t.out):
            if o < 0:
                continue
            step = -pl.out_t_stride if (pl.pair_flip >> (k & 3)) & 1 else pl.out_t_stride
            for st in range(2):
                seen[int(box[4]) + int(o) + st * step] += 1
    assert (seen == 1).all()
