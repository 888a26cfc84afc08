"""Parity of each HIP operator against a plain PyTorch fp32 CPU reference of the same op, through
the C-ABI (sd_op_* in include/sd_engine.h).  Inputs are rounded to fp16 first so both sides see
identical operands; tolerance = fp16 output rounding + fp32-accumulation order (stated per test)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def h(t):
    return t.half().cuda().contiguous()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, up, bias/rowadd/res
    (2, 16, 16, 64, 64, 3, 1, 0, True),
    (2, 16, 16, 128, 320, 3, 1, 0, True),      # Cout = 2.5 tiles of 128
    (1, 32, 32, 320, 320, 3, 1, 0, True),      # SD1.5 level-0 channel count
    (2, 16, 16, 64, 128, 3, 2, 0, True),       # Downsample2D
    (2, 8, 8, 128, 128, 3, 1, 1, True),        # Upsample2D: nearest 2x folded into the gather
    (2, 12, 20, 192, 64, 1, 1, 0, True),       # 1x1 conv / linear, ragged M
    (1, 7, 9, 64, 72, 3, 1, 0, False),         # ragged M and Cout % 64 != 0
    (3, 8, 8, 640, 1280, 1, 1, 0, False),
    (1, 8, 8, 1280, 1280, 3, 1, 0, True),      # small M, long K
    (1, 16, 16, 64, 4, 3, 1, 0, False),        # conv_out: Cout = 4 (scalar epilogue)
    (1, 16, 16, 64, 3, 3, 1, 0, False),        # VAE conv_out: Cout = 3
    (8, 8, 8, 1280, 1280, 3, 1, 0, True),      # 8 x 8 level of the C2 UNet: halo kernel, four whole images per tile, split over the slabs
    (4, 8, 8, 128, 320, 3, 1, 0, True),        # one tile of four images, no split (two slabs): fused epilogue with per-image row add
    (4, 8, 16, 192, 128, 3, 1, 0, True),       # 8 x 16 maps: two images per tile, 128-column form
    (8, 8, 8, 2560, 1280, 3, 1, 0, False),     # concat width of the up path at 8 x 8
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(engine_lib, case):
    N, H, W, Cin, Cout, k, stride, up, extras = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half()
    bias = torch.randn(Cout, generator=g) * 0.5 if extras else None
    rowadd = torch.randn(N, Cout, generator=g) * 0.5 if extras else None
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1 if k == 3 else 0)
    if rowadd is not None:
        ref = ref + rowadd[:, :, None, None]
    res = torch.randn(ref.shape, generator=g).half() if extras else None
    if res is not None:
        # engine rounds the conv result to fp16 before the residual add, like an fp16 torch graph
        ref = ref.half().float() + res.float()
    OH, OW = ref.shape[2], ref.shape[3]
    y = torch.empty(N, OH, OW, Cout, dtype=torch.float16, device="cuda")
    x_nhwc = h(x.permute(0, 2, 3, 1))
    res_nhwc = h(res.permute(0, 2, 3, 1)) if res is not None else None
    # keep every device tensor referenced until the call has completed
    wd = h(w)
    bd = bias.cuda() if bias is not None else None
    rd = rowadd.cuda().contiguous() if rowadd is not None else None
    rc = engine_lib.sd_op_conv2d(P(x_nhwc), P(wd), P(bd), P(rd), P(res_nhwc), P(y),
                                 N, H, W, Cin, Cout, k, stride, up, 0, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    got = y.permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 2e-3            # fp16 output rounding: ~5e-4 relative


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15])
@pytest.mark.parametrize("splits", [1, 3])
def test_conv2d_every_tile_variant(engine_lib, variant, splits):
    """Every LDS-DMA tile variant (and the split-K reduction) on a 3x3, a strided, an upsampled and a
    pointwise problem with ragged M / N; results must match regardless of the tile chosen."""
    engine_lib.sd_igemm_force(variant, splits)
    try:
        for case in [(2, 12, 20, 128, 320, 3, 1, 0, True), (1, 16, 16, 64, 192, 3, 2, 0, True),
                     (1, 9, 7, 128, 72, 3, 1, 1, True), (2, 17, 5, 256, 200, 1, 1, 0, True),
                     # shapes a 256-pixel patch tiles: the halo kernel (variant 10) takes them, the others run them as ordinary 3x3s
                     (2, 16, 16, 128, 200, 3, 1, 0, True), (1, 8, 32, 64, 320, 3, 1, 0, True), (1, 64, 64, 64, 72, 3, 1, 0, True),
                     (1, 32, 48, 64, 72, 3, 1, 0, True), (1, 8, 128, 128, 160, 3, 1, 0, True),    # 16- and 64-wide patches
                     # nearest-2x upsample fused into the gather, output images a 256-pixel patch tiles (halo kernels)
                     (1, 16, 16, 64, 200, 3, 1, 1, True), (2, 8, 32, 128, 160, 3, 1, 1, True), (1, 24, 8, 64, 128, 3, 1, 1, False)]:
            test_conv2d(engine_lib, case)
    finally:
        engine_lib.sd_igemm_force(-1, 0)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 64, 64, 320, 4), (1, 96, 80, 128, 3), (1, 13, 37, 64, 3), (3, 8, 8, 128, 4),
                                            (1, 40, 33, 192, 1)])
def test_conv3x3_small_cout_nchw(engine_lib, N, H, W, Cin, Cout):
    """conv_out (UNet 320 -> 4, VAE 128 -> 3): the dedicated HBM-bound kernel, NHWC in, NCHW out, against
    F.conv2d in fp32; ragged tiles and image borders included."""
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half()
    bias = torch.randn(Cout, generator=g) * 0.3
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    y = torch.empty(N, Cout, H, W, dtype=torch.float16, device="cuda")
    xd, wd, bd = h(x.permute(0, 2, 3, 1)), h(w), bias.cuda()
    rc = engine_lib.sd_op_conv3x3_small_cout(P(xd), P(wd), P(bd), P(y), N, H, W, Cin, Cout, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 2e-3


def test_conv_geglu(engine_lib):
    """Linear(C, 8C) + GEGLU epilogue: hidden * gelu(gate) with the 64-row interleaved packing."""
    g = torch.Generator().manual_seed(7)
    N, T, Cc = 2, 96, 64
    x = torch.randn(N * T, Cc, generator=g).half()
    w = (torch.randn(8 * Cc, Cc, generator=g) / Cc ** 0.5).half()
    b = torch.randn(8 * Cc, generator=g) * 0.2
    proj = (x.float() @ w.float().t() + b).half().float()
    hid, gate = proj.chunk(2, dim=-1)
    ref = hid * F.gelu(gate)
    y = torch.empty(N * T, 4 * Cc, dtype=torch.float16, device="cuda")
    xd, wd, bd = h(x), h(w), b.cuda()
    rc = engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, None, P(y), N, T, 1, Cc, 8 * Cc, 1, 1, 0, 1,
                                 stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 2e-3


@pytest.mark.parametrize("M,C", [(8192, 640), (2048, 1280), (1024, 640), (2560, 384)])
def test_geglu_persistent_projection(engine_lib, M, C):
    """geglu_persist_kernel (pgemm.hip): the FeedForward GEGLU projection of the 32 x 32 / 16 x 16 levels as one
    persistent block per CU whose DMA ring runs across tile boundaries -- the C2 shapes (5 and 2.5 tiles per block), a
    run of exactly two tiles per block, and ragged runs (10 M tiles x 24 N tiles over 250 blocks), against
    Linear -> chunk -> hidden * gelu(gate) in fp32.  (The LayerNorm-consumer form runs inside the UNet tests.)"""
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).half()
    w = (torch.randn(8 * C, C, generator=g) / C ** 0.5).half()
    b = torch.randn(8 * C, generator=g) * 0.2
    xd, wd, bd = h(x), h(w), b.cuda()
    proj = (xd.float() @ wd.float().t() + bd).half().float()
    hid, gate = proj.chunk(2, dim=-1)
    ref = hid * F.gelu(gate)
    y = torch.zeros(M, 4 * C, dtype=torch.float16, device="cuda")
    rc = engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, None, P(y), 1, M, 1, C, 8 * C, 1, 1, 0, 1, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    assert rel_l2(y, ref) < 2e-3
    # twice more on the same buffers: a race between the ring and a tile's reads would not repeat bit for bit
    for _ in range(2):
        y2 = torch.zeros_like(y)
        engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, None, P(y2), 1, M, 1, C, 8 * C, 1, 1, 0, 1, stream())
        torch.cuda.synchronize()
        assert torch.equal(y, y2)


@pytest.mark.parametrize("N,HW,C,silu,eps", [(2, 256, 64, 1, 1e-5), (2, 1024, 320, 1, 1e-5),
                                             (1, 64, 1920, 1, 1e-5), (3, 100, 128, 0, 1e-6),
                                             (1, 4096, 2560, 1, 1e-5), (2, 33, 960, 0, 1e-6),
                                             # single-pass (register-resident) form: every unit width / block shape
                                             (2, 64, 1280, 1, 1e-5), (2, 64, 2560, 1, 1e-5), (2, 256, 640, 1, 1e-5),
                                             (2, 256, 1920, 1, 1e-5), (2, 256, 2560, 0, 1e-5), (2, 1024, 640, 1, 1e-5),
                                             (1, 1024, 1920, 1, 1e-5), (2, 1024, 1280, 0, 1e-6), (2, 900, 512, 1, 1e-6),
                                             (1, 1, 320, 1, 1e-5), (2, 4096, 320, 1, 1e-5)])
def test_groupnorm(engine_lib, N, HW, C, silu, eps):
    g = torch.Generator().manual_seed(C + HW)
    x = (torch.randn(N, HW, C, generator=g) * 1.5 + 0.7).half()
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1)
    y = torch.empty(N, HW, C, dtype=torch.float16, device="cuda")
    xd, gd, bd = h(x), gamma.cuda(), beta.cuda()
    rc = engine_lib.sd_op_groupnorm(P(xd), P(gd), P(bd), P(y), N, HW, C, 32, eps, silu, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1.5e-3


@pytest.mark.parametrize("N,HW,C,offset", [(1, 262144, 128, 50.0),     # VAE 512 px level: 1 M elements per group
                                           (2, 4096, 320, 50.0),      # UNet 64 x 64 level (two-pass form)
                                           (2, 1024, 640, 50.0),      # single-kernel form
                                           (1, 16384, 512, -30.0)])
def test_groupnorm_large_mean_small_spread(engine_lib, N, HW, C, offset):
    """x = offset + 0.1 * randn.  With single-pass E[x^2] - mean^2 in fp32 the variance (0.01)
    drowns in the cancellation (mean^2 = 2500) and rstd goes wrong by orders of magnitude; the kernels
    keep shifted sums / (mean, M2) summaries merged with Chan's formula.  fp32 F.group_norm is the checker.
    The input's own fp16 grid (spacing 0.03 at 50) is part of x on both sides."""
    g = torch.Generator().manual_seed(HW + C)
    x = (offset + 0.1 * torch.randn(N, HW, C, generator=g)).half()
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1)
    y = torch.empty(N, HW, C, dtype=torch.float16, device="cuda")
    xd, gd, bd = h(x), gamma.cuda(), beta.cuda()
    rc = engine_lib.sd_op_groupnorm(P(xd), P(gd), P(bd), P(y), N, HW, C, 32, 1e-6, 0, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1.5e-3


CONV_GN_CASES = [
    # N, H, W, Cin, Cout, k, stride, up, residual, expect the epilogue path
    # `expect`: 1 = the tuned table gives this shape a tile whose epilogue can leave the summaries (C2's own
    # shapes), None = whichever tile / split-K the heuristic picks, the numbers must not depend on it
    (8, 64, 64, 320, 320, 3, 1, 0, True, 1),      # UNet level 0 resnet conv2: the 160-column halo kernel in its
                                                  # rows-as-loop form (the unrolled one has no registers left for them)
    (8, 64, 64, 320, 320, 1, 1, 0, True, 1),      # Transformer2D proj_out (+residual) -> conv_norm_out / next block
    (2, 64, 64, 320, 320, 3, 1, 0, True, None),
    (2, 64, 64, 64, 320, 1, 1, 0, False, None),   # conv_in as the im2col GEMM (pointwise)
    (4, 128, 128, 256, 256, 3, 1, 0, True, None), # VAE level, cpg = 8
    (1, 64, 64, 128, 128, 3, 1, 1, False, None),  # VAE upsample conv (2x nearest in the gather), cpg = 4
    (1, 128, 128, 128, 512, 3, 2, 0, False, None),   # stride 2 (encoder / UNet downsample), cpg = 16
    (2, 32, 32, 640, 640, 3, 1, 0, True, None),   # small map: the single-kernel GroupNorm needs no summaries
    (1, 40, 40, 64, 320, 3, 1, 0, False, None),   # 1600 pixels per image
    (8, 16, 16, 1280, 1280, 3, 1, 0, True, 1),    # UNet 16 x 16 level: split-K, summaries from the reduction kernel
    (8, 8, 8, 1280, 1280, 3, 1, 0, True, 1),      # 8 x 8 level: split-K 8, 16-pixel slabs (4 per image)
    (8, 32, 32, 960, 640, 3, 1, 0, False, None),  # 32 x 32 level, halo kernel split over the slabs
    (2, 16, 16, 1280, 1280, 1, 1, 0, True, None), # proj_out at 16 x 16
]


@pytest.mark.parametrize("case", CONV_GN_CASES)
def test_conv_groupnorm_statistics_from_the_conv_epilogue(engine_lib, case):
    """conv -> GroupNorm(+SiLU) with the GroupNorm summaries written by the convolution's epilogue
    (IGemmParams::gnstat_out) against conv2d + group_norm in fp32; also checks which path ran."""
    N, H, W, Cin, Cout, k, stride, up, with_res, expect = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half()
    bias = torch.randn(Cout, generator=g) * 0.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cout, generator=g)
    beta = 0.2 * torch.randn(Cout, generator=g)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    conv = F.conv2d(xin, w.float(), bias, stride=stride, padding=1 if k == 3 else 0)
    res = torch.randn(conv.shape, generator=g).half() if with_res else None
    conv = conv.half().float()
    if res is not None:
        conv = (conv + res.float()).half().float()
    ref = F.silu(F.group_norm(conv, 32, gamma, beta, 1e-5))
    OH, OW = conv.shape[2], conv.shape[3]
    yc = torch.empty(N, OH, OW, Cout, dtype=torch.float16, device="cuda")
    yg = torch.empty_like(yc)
    fused = C.c_int(-1)
    xd = h(x.permute(0, 2, 3, 1))
    rd = h(res.permute(0, 2, 3, 1)) if res is not None else None
    wd, bd, gd, betad = h(w), bias.cuda(), gamma.cuda(), beta.cuda()     # keep the device copies alive across the call
    rc = engine_lib.sd_op_conv2d_groupnorm(P(xd), P(wd), P(bd), None, P(rd), P(yc), P(gd), P(betad), P(yg), N, H, W,
                                           Cin, Cout, k, stride, up, 32, 1e-5, 1, C.byref(fused), stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    print("conv->GN", case, "summaries from the conv epilogue:", fused.value)
    if expect is not None:
        assert fused.value == expect
    assert rel_l2(yc.permute(0, 3, 1, 2), conv) < 2e-3
    assert rel_l2(yg.permute(0, 3, 1, 2), ref) < 3e-3


@pytest.mark.parametrize("rows,C", [(300, 320), (1000, 640), (77, 1280), (5, 64)])
def test_layernorm(engine_lib, rows, C):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).half()
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    y = torch.empty(rows, C, dtype=torch.float16, device="cuda")
    xd, gd, bd = h(x), gamma.cuda(), beta.cuda()
    rc = engine_lib.sd_op_layernorm(P(xd), P(gd), P(bd), P(y), rows, C, 1e-5, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1.5e-3


ATTN_CASES = [
    # B, Tq, Tk, heads, d
    (2, 256, 256, 8, 40),      # SD1.5 level 0 head dim (padded 40 -> 64 / 48)
    (1, 1024, 1024, 8, 80),
    (2, 256, 256, 8, 160),
    (2, 64, 64, 8, 160),       # 8x8 latents: one partial query block
    (2, 256, 77, 8, 40),       # cross attention, ragged keys
    (1, 100, 77, 5, 64),       # SDXL head dim, ragged queries
    (2, 128, 77, 4, 32),       # test-config head dim
    (1, 320, 320, 1, 512),     # VAE mid-block attention (single head over 512 channels)
    (1, 192, 192, 1, 128),     # tiny-VAE mid-block
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention(engine_lib, case):
    B, Tq, Tk, H, d = case
    g = torch.Generator().manual_seed(sum(case))
    q = torch.randn(B, Tq, H * d, generator=g).half()
    k = torch.randn(B, Tk, H * d, generator=g).half()
    v = torch.randn(B, Tk, H * d, generator=g).half()
    ref = F.scaled_dot_product_attention(q.float().view(B, Tq, H, d).transpose(1, 2),
                                         k.float().view(B, Tk, H, d).transpose(1, 2),
                                         v.float().view(B, Tk, H, d).transpose(1, 2))
    ref = ref.transpose(1, 2).reshape(B, Tq, H * d)
    out = torch.empty(B, Tq, H * d, dtype=torch.float16, device="cuda")
    qd, kd, vd = h(q), h(k), h(v)
    rc = engine_lib.sd_op_attention(P(qd), P(kd), P(vd), P(out), B, Tq, Tk, H, d, H * d, H * d, H * d, H * d,
                                    stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    # P is rounded to fp16 before the PV product (like fp16 SDPA): ~1e-3 relative
    assert rel_l2(out, ref) < 3e-3


def test_attention_online_softmax_rescale(engine_lib):
    """Force the running-max rescale branch: one key far above the rest in a LATER key tile."""
    B, T, H, d = 1, 256, 2, 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, T, H * d, generator=g).half()
    k = (torch.randn(B, T, H * d, generator=g) * 0.3).half()
    v = torch.randn(B, T, H * d, generator=g).half()
    k[:, 200] = q[:, 17] * 2.0        # spikes the score of query 17 at key 200 (4th tile)
    ref = F.scaled_dot_product_attention(q.float().view(B, T, H, d).transpose(1, 2),
                                         k.float().view(B, T, H, d).transpose(1, 2),
                                         v.float().view(B, T, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, T, H * d)
    out = torch.empty(B, T, H * d, dtype=torch.float16, device="cuda")
    qd, kd, vd = h(q), h(k), h(v)
    rc = engine_lib.sd_op_attention(P(qd), P(kd), P(vd), P(out), B, T, T, H, d, H * d, H * d, H * d, H * d, stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 3e-3
    assert (out[0, 17].float().cpu() - ref[0, 17]).abs().max() < 2e-2


@pytest.mark.parametrize("case", ATTN_CASES[:8] + [(1, 4096, 4096, 2, 40), (1, 300, 300, 2, 64)])
@pytest.mark.parametrize("spike", [False, True])
def test_attention_prescaled_queries(engine_lib, case, spike):
    """The `prescaled` path the UNet uses: q arrives multiplied by log2(e)/sqrt(d) and the running
    reference is subtracted through the MFMA accumulators.  Checked against SDPA of the SAME fp16
    queries (un-scaled on the host in fp32), with and without a late score spike that forces the
    reference to move, and with all-negative scores in the first tile (reference set downwards)."""
    B, Tq, Tk, H, d = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    c = 1.4426950408889634 / d ** 0.5
    qs = (torch.randn(B, Tq, H * d, generator=g) * c).half()          # what the scaled projection would emit
    k = (torch.randn(B, Tk, H * d, generator=g) * (0.3 if spike else 1.0)).half()
    v = torch.randn(B, Tk, H * d, generator=g).half()
    if spike and Tk > 70:
        k[:, Tk - 3] = (qs[:, min(17, Tq - 1)].float() / c * 2.0).half()   # big score in the last key tile
        k[:, :64] -= 0                                                        # (first tile stays ordinary)
    q_unscaled = qs.float() / c
    ref = F.scaled_dot_product_attention(q_unscaled.view(B, Tq, H, d).transpose(1, 2),
                                         k.float().view(B, Tk, H, d).transpose(1, 2),
                                         v.float().view(B, Tk, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, Tq, H * d)
    out = torch.empty(B, Tq, H * d, dtype=torch.float16, device="cuda")
    qd, kd, vd = h(qs), h(k), h(v)
    rc = engine_lib.sd_op_attention_ex(P(qd), P(kd), P(vd), P(out), B, Tq, Tk, H, d, H * d, H * d, H * d, H * d, 0, 1,
                                       stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) < 3e-3


def test_attention_prescaled_very_negative_scores(engine_lib):
    """All scores of a row far below zero (log2 domain): the first tile must pull the reference down,
    otherwise every probability underflows and the row divides by zero."""
    B, T, H, d = 1, 128, 1, 64
    g = torch.Generator().manual_seed(11)
    c = 1.4426950408889634 / d ** 0.5
    base = torch.randn(1, 1, d, generator=g)
    q = (base * 6.0).expand(B, T, d).clone()
    k = (-base * 6.0 + 0.1 * torch.randn(B, T, d, generator=g))      # q.k ~ -36 * |base|^2 * ... : hugely negative
    v = torch.randn(B, T, d, generator=g)
    qs = (q * c).half()
    ref = F.scaled_dot_product_attention((qs.float() / c).view(B, T, H, d).transpose(1, 2),
                                         k.half().float().view(B, T, H, d).transpose(1, 2),
                                         v.half().float().view(B, T, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, T, d)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    qd, kd, vd = h(qs), h(k.half()), h(v.half())
    rc = engine_lib.sd_op_attention_ex(P(qd), P(kd), P(vd), P(out), B, T, T, H, d, d, d, d, d, 0, 1, stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(8, 64, 64, 320, 0, False), (8, 64, 64, 960, 0, False), (3, 64, 64, 320, 0, True),
                                  (2, 64, 64, 640, 0, True), (1, 32, 32, 320, 0, False), (8, 64, 64, 2560, 1, False),
                                  (5, 64, 48, 2560, 1, False), (2, 64, 64, 256, 1, False)])
def test_weight_stationary_gemm(engine_lib, case):
    """K = 320 pointwise problems with M % 128 == 0 go to wsgemm_kernel (persistent, weight-stationary; igemm2
    variants 13 / 14): plain, residual and GEGLU epilogues against torch fp32, runs of one to twenty tiles per block
    and W reloads inside a block (the LayerNorm-consumer and row-statistics forms run inside the UNet:
    tests/test_fullsize_gpu.py)."""
    N, H, W, Cout, geglu, res = case
    Cin = 320
    g = torch.Generator().manual_seed(Cout + N)
    x = torch.randn(N, H, W, Cin, generator=g).half()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    oc = Cout // 2 if geglu else Cout
    r = torch.randn(N, H, W, oc, generator=g).half() if res else None
    ref = F.linear(x.float(), w.float().view(Cout, Cin), b)
    if geglu:
        hh, gg = ref.chunk(2, dim=-1)
        ref = hh * F.gelu(gg)
    if res:
        ref = ref.half().float() + r.float()
    xd, wd, bd = h(x), h(w), b.cuda()
    rd = h(r) if res else None
    y = torch.zeros(N, H, W, oc, dtype=torch.float16, device="cuda")
    engine_lib.sd_igemm_force(14 if geglu else 13, 1)      # (launches with a residual are not routed there by default)
    try:
        rc = engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, P(rd) if res else None, P(y), N, H, W, Cin, Cout, 1, 1, 0,
                                     geglu, stream())
    finally:
        engine_lib.sd_igemm_force(-1, 0)
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 2e-3


def test_wsgemm_residual_loads_need_no_range_check(engine_lib, monkeypatch):
    """Round 2 saw HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in wsgemm_kernel<160, false, false, true, false> while its
    residual prefetch was an inline-asm global_load; the range-checked raw_buffer_load that replaced it could have been
    hiding a prefetch past the last row of a run.  It is not: with the residual's descriptor opened up
    (no range check: an over-run would read whatever lies behind the tensor -- a NaN guard here) the outputs are bit for
    bit those of the checked run, on the shape that faulted (8 x 64 x 64, 320 -> 320, 256 blocks x 4 runs) and on ragged
    run lengths.  (tests/test_wsgemm_indexing.py walks the same index arithmetic on the host.)"""
    for (N, H, W, Cout) in [(8, 64, 64, 320), (3, 64, 64, 320), (2, 64, 64, 640), (5, 64, 48, 960)]:
        Cin = 320
        M = N * H * W
        g = torch.Generator().manual_seed(Cout + N)
        x = torch.randn(M, Cin, generator=g).half().cuda()
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).half().cuda()
        b = torch.randn(Cout, generator=g).cuda()
        guard = torch.full((M + 4096, Cout), float("nan"), dtype=torch.float16, device="cuda")
        guard[:M] = torch.randn(M, Cout, generator=g).half().cuda()
        r = guard[:M]
        outs = []
        for unchecked in (False, True):
            if unchecked:
                monkeypatch.setenv("SD_WS_RES_UNCHECKED", "1")
            else:
                monkeypatch.delenv("SD_WS_RES_UNCHECKED", raising=False)
            y = torch.zeros(M, Cout, dtype=torch.float16, device="cuda")
            engine_lib.sd_igemm_force(13, 1)
            try:
                rc = engine_lib.sd_op_conv2d(P(x), P(w), P(b), None, P(r), P(y), N, H, W, Cin, Cout, 1, 1, 0, 0, stream())
            finally:
                engine_lib.sd_igemm_force(-1, 0)
            assert rc == 0, engine_lib.sd_last_error()
            torch.cuda.synchronize()
            outs.append(y)
        monkeypatch.delenv("SD_WS_RES_UNCHECKED", raising=False)
        assert torch.isfinite(outs[1].float()).all()
        assert torch.equal(outs[0], outs[1])
        ref = (F.linear(x.float(), w.float().view(Cout, Cin), b).half().float() + r.float())
        assert rel_l2(outs[1], ref) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dim,flip,shift", [(320, 1, 0.0), (256, 1, 0.0), (320, 0, 1.0), (64, 0, 0.0)])
def test_timestep_sinusoid(engine_lib, dim, flip, shift):
    """`Timesteps` / get_timestep_embedding (time_proj of the UNet, add_time_proj of SDXL) against the oracle's
    restatement: integer, fractional (Euler / DPM sigmas map to fractional timesteps) and large arguments."""
    from oracle.unet_ref import timestep_sinusoid
    t = torch.tensor([0.0, 1.0, 17.0, 250.5, 501.0, 980.9999, 999.0, 1024.0])
    ref = timestep_sinusoid(t, dim, bool(flip), shift)
    td = t.cuda()
    out = torch.zeros(len(t), dim, dtype=torch.float32, device="cuda")
    rc = engine_lib.sd_op_timestep_sinusoid(P(td), P(out), len(t), dim, flip, shift, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    # the arguments reach ~1e3 rad: fp32 range reduction on both sides, a few 1e-4 absolute
    assert (out.cpu() - ref).abs().max() < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,n_out,silu_in,silu_out", [(8, 320, 1280, 0, 1), (8, 1280, 1280, 0, 0), (2, 2816, 1280, 0, 1),
                                                      (8, 1280, 320 * 22, 1, 0), (1, 64, 40, 1, 1),
                                                      (16, 1280, 20160, 1, 0), (3, 320, 1280, 0, 1), (8, 96, 48, 0, 0)])
def test_small_linear(engine_lib, B, K, n_out, silu_in, silu_out):
    """The time-embedding MLP linears (TimestepEmbedding.linear_1 / linear_2, SDXL add_embedding, the stacked
    time_emb_proj of every resnet): fp32 activations, fp16 weights, optional SiLU before / after.  K <= 1280 with
    n_out % 16 == 0 runs skinny_linear_kernel (MFMA, x split hi + lo so that it stays fp32-exact), the rest the GEMV kernel."""
    g = torch.Generator().manual_seed(K + n_out)
    x = torch.randn(B, K, generator=g)
    w = (torch.randn(n_out, K, generator=g) / K ** 0.5).half()
    b = torch.randn(n_out, generator=g)
    xi = F.silu(x) if silu_in else x
    ref = F.linear(xi, w.float(), b)
    if silu_out:
        ref = F.silu(ref)
    xd, wd, bd = x.cuda(), h(w), b.cuda()
    y = torch.zeros(B, n_out, dtype=torch.float32, device="cuda")
    rc = engine_lib.sd_op_small_linear(P(xd), P(wd), P(bd), P(y), B, K, n_out, silu_in, silu_out, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1e-4


GN_CONV_CASES = [
    # N, H, W, Cin, Cout, groups, silu, extras, x_scale, x_shift
    (2, 16, 16, 64, 64, 32, 1, True, 1.0, 0.0),        # one slab, 16 x 16 patch
    (2, 32, 32, 128, 320, 32, 1, True, 1.0, 0.0),      # two slabs, 32-wide patch, Cout = 2 x 160
    (1, 64, 64, 320, 320, 32, 1, True, 2.0, 0.5),      # SD1.5 level 0: five slabs, 64-wide patch, cpg = 10
    (2, 16, 16, 640, 128, 32, 1, False, 1.0, 0.0),     # ten slabs, cpg = 20, split over the slabs (few tiles)
    (3, 16, 32, 192, 256, 32, 0, True, 1.0, -1.0),     # no SiLU, cpg = 6 (chunks straddle groups), 128-column form
    (1, 32, 32, 960, 320, 32, 1, True, 1.0, 0.0),      # concat width of the up path: cpg = 30
    (1, 32, 32, 128, 128, 32, 1, False, 0.1, 50.0),    # |mean| >> std: the affine runs in fp32
    (2, 8, 8, 128, 128, 32, 1, True, 1.0, 0.0),        # 8 x 8 maps: no halo patch -> GroupNorm kernel + conv (fused = 0)
]


@pytest.mark.parametrize("case", GN_CONV_CASES)
def test_groupnorm_conv2d(engine_lib, case):
    """GroupNorm -> SiLU -> 3x3 conv with the norm applied inside the convolution (conv3x3_halo_kernel<BN, true>:
    halo tiles normalised in LDS; borders must stay zero = the conv pads the NORMALISED tensor) against
    F.group_norm -> F.silu -> F.conv2d in fp32."""
    N, H, W, Cin, Cout, G, silu, extras, xs, xsh = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:6]) + 5)
    x = (torch.randn(N, Cin, H, W, generator=g) * xs + xsh + 0.3 * torch.randn(1, Cin, 1, 1, generator=g)).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half()
    gamma = 1.0 + 0.3 * torch.randn(Cin, generator=g)
    beta = 0.3 * torch.randn(Cin, generator=g)
    bias = torch.randn(Cout, generator=g) * 0.5 if extras else None
    rowadd = torch.randn(N, Cout, generator=g) * 0.5 if extras else None
    hn = F.group_norm(x.float(), G, gamma, beta, 1e-5)
    if silu:
        hn = F.silu(hn)
    ref = F.conv2d(hn, w.float(), bias, padding=1)
    if rowadd is not None:
        ref = ref + rowadd[:, :, None, None]
    res = torch.randn(ref.shape, generator=g).half() if extras else None
    if res is not None:
        ref = ref.half().float() + res.float()
    y = torch.empty(N, H, W, Cout, dtype=torch.float16, device="cuda")
    xd = h(x.permute(0, 2, 3, 1))
    wd = h(w)
    bd = bias.cuda() if bias is not None else None
    rd = rowadd.cuda().contiguous() if rowadd is not None else None
    resd = h(res.permute(0, 2, 3, 1)) if res is not None else None
    gd, btd = gamma.cuda(), beta.cuda()
    fused = C.c_int(-1)
    rc = engine_lib.sd_op_groupnorm_conv2d(P(xd), P(gd), P(btd), G, 1e-5, silu, P(wd), P(bd), P(rd), P(resd), P(y), N, H, W,
                                           Cin, Cout, 3, 0, None, C.byref(fused), stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert fused.value == (0 if H == 8 else 1)
    out = y.float().cpu().permute(0, 3, 1, 2)
    # the normalised activations are rounded to fp16 once (as the separate kernel did): conv-level tolerance
    assert rel_l2(out, ref) < 3e-3, rel_l2(out, ref)
    # borders: the outermost output ring sees the zero padding of the normalised tensor
    ring = torch.ones(H, W, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    assert rel_l2(out[:, :, ring], ref[:, :, ring]) < 4e-3


@pytest.mark.parametrize("M,C,scale", [(8192, 320, 1.0), (32768, 320, 1.0), (8320, 320, 3.0), (4096, 640, 1.0)])
def test_ffn_geglu_fused(engine_lib, M, C, scale):
    """x + GEGLU(LN(x) W1 + b1) W2 + b2 against LayerNorm -> Linear -> chunk -> h * gelu(g) -> Linear -> + x in fp32.
    C = 320: ffn_fused_kernel (the hidden tensor never leaves the CU; 64 / 256 / 65 blocks); C = 640: the two-GEMM form
    through the same entry (fused = 0)."""
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * scale + 0.2 * torch.randn(1, C, generator=g)).half()
    w1 = (torch.randn(8 * C, C, generator=g) / C ** 0.5).half()
    b1 = torch.randn(8 * C, generator=g) * 0.2
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).half()
    b2 = torch.randn(C, generator=g) * 0.2
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    xd, w1d, w2d = h(x), h(w1), h(w2)
    b1d, b2d, gd, bd = b1.cuda(), b2.cuda(), gamma.cuda(), beta.cuda()
    with torch.no_grad():
        xf = xd.float()
        proj = F.linear(F.layer_norm(xf, (C,), gd, bd, 1e-5), w1d.float(), b1d)
        hid, gate = proj.chunk(2, dim=-1)
        ref = xf + F.linear(hid * F.gelu(gate), w2d.float(), b2d)
    y = torch.zeros(M, C, dtype=torch.float16, device="cuda")
    fused = C.c_int(-1) if False else __import__("ctypes").c_int(-1)
    rc = engine_lib.sd_op_ffn_geglu(P(xd), P(gd), P(bd), 1e-5, P(w1d), P(b1d), P(w2d), P(b2d), P(y), M, C, 0, None,
                                    __import__("ctypes").byref(fused), stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert fused.value == (1 if C == 320 else 0)
    assert torch.isfinite(y.float()).all()
    assert rel_l2(y, ref) < 3e-3, rel_l2(y, ref)
    y2 = torch.zeros_like(y)           # again on the same buffers: a race in the slab ring would not repeat bit for bit
    engine_lib.sd_op_ffn_geglu(P(xd), P(gd), P(bd), 1e-5, P(w1d), P(b1d), P(w2d), P(b2d), P(y2), M, C, 0, None, None, stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("N,Cin,H,W", [(8, 4, 64, 64), (2, 4, 16, 16), (1, 4, 128, 128), (2, 7, 16, 24)])
def test_unet_conv_in_one_launch(engine_lib, N, Cin, H, W):
    """conv_head_kernel: conv_in straight from the NCHW latents (im2col tile in LDS), output NHWC, plus the GroupNorm
    summaries (mean, M2) of every 128-pixel tile x group of the STORED fp16 output."""
    import ctypes
    Cout, G = 320, 32
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).half()
    b = torch.randn(Cout, generator=g) * 0.3
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    y = torch.zeros(N, H, W, Cout, dtype=torch.float16, device="cuda")
    S = H * W // 128
    summ = torch.zeros(N, S, G, 2, dtype=torch.float32, device="cuda")
    xd, wd, bd = h(x), h(w), b.cuda()
    rc = engine_lib.sd_op_unet_conv_in(P(xd), P(wd), P(bd), P(y), P(summ), G, N, Cin, H, W, Cout, 0, None, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    out = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(out, ref) < 1e-3, rel_l2(out, ref)
    ring = torch.ones(H, W, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    assert rel_l2(out[:, :, ring], ref[:, :, ring]) < 1e-3
    # summaries of the stored values, tile = 128 consecutive pixels of one image
    t = y.float().cpu().reshape(N, S, 128, G, Cout // G)
    mean = t.mean(dim=(2, 4))
    m2 = ((t - mean[:, :, None, :, None]) ** 2).sum(dim=(2, 4))
    got = summ.cpu()
    assert torch.allclose(got[..., 0], mean, atol=2e-5, rtol=1e-4)
    assert torch.allclose(got[..., 1], m2, atol=1e-3, rtol=1e-4)


def test_unet_conv_in_refuses_other_shapes(engine_lib):
    x = torch.zeros(1, 9, 16, 16, dtype=torch.float16, device="cuda")          # inpainting's 9 channels: K = 81 > 64
    w = torch.zeros(320, 9, 3, 3, dtype=torch.float16, device="cuda")
    b = torch.zeros(320, device="cuda")
    y = torch.zeros(1, 16, 16, 320, dtype=torch.float16, device="cuda")
    assert engine_lib.sd_op_unet_conv_in(P(x), P(w), P(b), P(y), None, 0, 1, 9, 16, 16, 320, 0, None, stream()) != 0
    assert b"one-launch" in engine_lib.sd_last_error()


@pytest.mark.parametrize("N,H,W,Cout,silu", [(8, 64, 64, 4, 1), (2, 16, 16, 4, 1), (1, 128, 128, 4, 1), (2, 8, 32, 3, 0)])
def test_unet_conv_out_one_launch(engine_lib, N, H, W, Cout, silu):
    """conv_tail_kernel: GroupNorm + SiLU + 3x3 convolution to <= 4 channels + NCHW in one launch against
    group_norm -> silu -> conv2d in fp32 (the normalised activations are rounded to fp16 once, as the separate kernels do)."""
    C_, G = 320, 32
    g = torch.Generator().manual_seed(N * H + Cout)
    x = (torch.randn(N, C_, H, W, generator=g) * (1 + torch.rand(1, C_, 1, 1, generator=g)) + torch.randn(1, C_, 1, 1, generator=g)).half()
    gamma = 1 + 0.2 * torch.randn(C_, generator=g)
    beta = 0.2 * torch.randn(C_, generator=g)
    w = (torch.randn(Cout, C_, 3, 3, generator=g) / (9 * C_) ** 0.5).half()
    b = torch.randn(Cout, generator=g) * 0.3
    hn = F.group_norm(x.float(), G, gamma, beta, 1e-5)
    if silu:
        hn = F.silu(hn)
    ref = F.conv2d(hn, w.float(), b, padding=1)
    y = torch.zeros(N, Cout, H, W, dtype=torch.float16, device="cuda")
    xd, wd = h(x.permute(0, 2, 3, 1)), h(w)
    gd, btd, bd = gamma.cuda(), beta.cuda(), b.cuda()
    rc = engine_lib.sd_op_unet_conv_out(P(xd), P(gd), P(btd), G, 1e-5, silu, P(wd), P(bd), P(y), N, H, W, C_, Cout, 0, None, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    out = y.float().cpu()
    assert rel_l2(out, ref) < 3e-3, rel_l2(out, ref)
    ring = torch.ones(H, W, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    assert rel_l2(out[:, :, ring], ref[:, :, ring]) < 4e-3


@pytest.mark.parametrize("N,HW,Ca,Cb", [(8, 4096, 640, 320), (2, 4096, 320, 320), (2, 1024, 1280, 640), (2, 1024, 640, 640),
                                        (3, 1024, 640, 320), (2, 1280, 640, 320)])
def test_groupnorm_of_a_concatenation_from_the_halves_summaries(engine_lib, N, HW, Ca, Cb):
    """The up blocks' norm1 over torch.cat([hidden, skip]): per-half summaries (the hidden half over sub-groups of width
    gcd(C / 32, Ca): 960 = 640 + 320 has a group that straddles the seam) merged by gn_cat_finalize_kernel, then the apply
    pass -- against F.group_norm of the concatenated tensor.  The last case has a ragged last slab."""
    G = 32
    Cc = Ca + Cb
    g = torch.Generator().manual_seed(HW + Ca)
    x = torch.randn(N, HW, Cc, generator=g) * (0.5 + torch.rand(1, 1, Cc, generator=g) * 2) + torch.randn(1, 1, Cc, generator=g) * 3
    x[..., Ca:] += 5.0                                      # the halves sit at different levels: a straddling group sees both
    x = x.half()
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.2 * torch.randn(Cc, generator=g)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1)
    xd, gd, bd = h(x), gamma.cuda(), beta.cuda()
    y = torch.zeros_like(xd)
    rc = engine_lib.sd_op_groupnorm_concat(P(xd), Ca, Cb, P(gd), P(bd), P(y), N, HW, G, 1e-5, 1, stream())
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)
    assert (y.float().cpu() - ref).abs().max() < 2e-2


@pytest.mark.parametrize("M,Cin,Cout,bias,res", [(2048, 1280, 1280, True, True), (8192, 640, 640, False, False), (2048, 5120, 1280, True, True),
                                                 (1000, 256, 200, True, True), (130, 320, 88, True, False), (512, 1280, 3840, False, False)])
def test_pointwise_gemm_activations_through_registers(engine_lib, M, Cin, Cout, bias, res):
    """igemm3_kernel (variant 18): the 128 x 80 pointwise tile whose activation fragments are fetched straight into registers
    (only the weights use the LDS-DMA ring) against F.linear in fp32; ragged M and N, with and without bias / residual."""
    g = torch.Generator().manual_seed(M + Cout)
    x = torch.randn(M, Cin, generator=g).half()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).half()
    b = torch.randn(Cout, generator=g) * 0.5 if bias else None
    r = torch.randn(M, Cout, generator=g).half() if res else None
    ref = F.linear(x.float(), w.float(), b)
    if r is not None:
        ref = ref.half().float() + r.float()
    y = torch.zeros(M, Cout, dtype=torch.float16, device="cuda")
    xd, wd = h(x), h(w)
    bd = b.cuda() if b is not None else None
    rd = h(r) if r is not None else None
    engine_lib.sd_igemm_force(18, 1)
    try:
        rc = engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, P(rd), P(y), 1, M, 1, Cin, Cout, 1, 1, 0, 0, stream())
        assert rc == 0, engine_lib.sd_last_error()
        torch.cuda.synchronize()
        y2 = torch.zeros_like(y)                 # twice on the same buffers: a race in the fragment ring would not repeat bit for bit
        engine_lib.sd_op_conv2d(P(xd), P(wd), P(bd), None, P(rd), P(y2), 1, M, 1, Cin, Cout, 1, 1, 0, 0, stream())
        torch.cuda.synchronize()
    finally:
        engine_lib.sd_igemm_force(-1, 0)
    assert rel_l2(y, ref) < 2e-3, rel_l2(y, ref)
    assert torch.equal(y, y2)
