"""Parity of every C-ABI entry point against the oracle / a plain PyTorch-CPU fp32 reference on seeded inputs.
Each case runs on 'emu' (hipemu CPU build of the same HIP sources; CPU tier) and on 'hip' (real MI355X; -m gpu)."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BOTH, HIP_ONLY, WITH_LATE_DMA
from emu_util import bf16_round, empty_nhwc, from_nhwc, h16_round, to_nhwc, bf16_bits_to_f32
from oracle import cpu_ops as O
from oracle import graph_ref as G
from oracle import input_ref as IR
from rangedet_amd import lib as R
from rangedet_amd import synth
from rangedet_amd.runtime import bn_affine

F32, BF16, F16 = R.RD_F32, R.RD_BF16, R.RD_F16
H16 = R.H16


def _gran(dt):
    return 16 if dt in H16 else 8


def _ulp(dt):
    """Half-ulp rounding of the 16-bit types, with margin: bf16 2^-9 -> 2^-8, fp16 2^-12 -> 2^-11."""
    return 2 ** -8 if dt == BF16 else 2 ** -11


def _tol(dt, ref):
    # f32: accumulation-order noise only.  16-bit: inputs are pre-rounded to the type so the only error is the output rounding
    # (half an ulp: 2^-9 relative for bf16, 2^-12 for fp16) plus fp32 accumulation.
    return 2e-5 * max(1.0, float(np.abs(ref).max())) if dt == F32 else _ulp(dt) * max(1.0, float(np.abs(ref).max())) + (2e-5 if dt == F16 else 0)


def run_conv(be, dt, B, H, W, cin, cout, k, stride, flags, cs_in=None, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    Wout = (W + 2 * (k // 2) - k) // stride + 1
    res = rng.standard_normal((B, cout, H, Wout)).astype(np.float32)
    x, w, res = h16_round(x, dt), h16_round(w, dt), h16_round(res, dt)
    g = _gran(dt)
    cs = cs_in or -(-cin // g) * g
    L = be.lib
    xin, rin = be.up(to_nhwc(x, dt, cstride=cs)), be.up(to_nhwc(res, dt))
    y = be.empty(B * H * Wout * cout * 4)
    wp, dsc, dsh = be.up(L.pack_conv_weight(w, dt)), be.up(sc), be.up(sh)
    L.call("rd_conv2d_bn_act", be.ptr(xin), cs, 0, be.ptr(wp), be.ptr(dsc), be.ptr(dsh), be.ptr(rin), cout, 0, be.ptr(y),
           cout, 0, B, H, W, cin, cout, k, k, stride, flags, dt, be.stream)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=(1, stride), padding=k // 2).numpy()
    ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    if flags & R.RD_RELU_PRE:
        ref = np.maximum(ref, 0)
    if flags & R.RD_ADD:
        ref = ref + res
    if flags & R.RD_RELU_POST:
        ref = np.maximum(ref, 0)
    raw = be.down(y, np.uint16 if dt in H16 else np.float32, (B, H, Wout, cout))
    got = from_nhwc(raw, dt, cout)
    assert np.abs(got - ref).max() <= _tol(dt, ref), (np.abs(got - ref).max(), _tol(dt, ref))


CONV_CASES = [
    # dt, B, H, W, cin, cout, k, stride, flags
    (F32, 1, 5, 70, 8, 64, 3, 1, 4),       # first layer (8 input channels), partial column tile, H not multiple of RO
    (F32, 1, 3, 40, 64, 128, 3, 1, 6),     # residual add + relu, cout 128
    (BF16, 1, 4, 70, 16, 64, 3, 1, 4),
    (BF16, 2, 3, 33, 128, 128, 3, 2, 6),   # stride (1,2), two k-chunks, batch 2
    (BF16, 1, 2, 20, 72, 128, 3, 1, 4),    # head level 0: 72 channels in an 80-wide buffer (partial k-chunk)
    (F32, 1, 2, 21, 64, 128, 1, 2, 0),     # projection shortcut 1x1 stride 2, no activation
    (F32, 1, 2, 64, 128, 64, 3, 1, 1),     # relu before (no) add
    # persistent 3x3 kernel (k_conv3.h): several tiles per workgroup (emu: 4 "CUs"), column tiles 126 wide, batch, residual
    (BF16, 2, 9, 300, 64, 128, 3, 1, 6),
    (BF16, 1, 6, 127, 128, 64, 3, 1, 5),   # W = 127: a second column tile of one pixel; relu before + relu after
    (BF16, 3, 4, 126, 8, 64, 3, 1, 4),     # one 32-channel unit per tile (every unit starts a new tile)
    (BF16, 2, 5, 260, 64, 64, 3, 2, 6),    # stride 2 = stride 1 with the even columns stored
    (BF16, 1, 4, 131, 128, 128, 3, 2, 4),  # stride 2, odd width
    (BF16, 1, 27, 200, 64, 64, 3, 1, 6),   # 4 x 4 tiles: the middle ones are interior units (halo pieces with a uniform base)
    (BF16, 1, 26, 190, 128, 128, 3, 1, 4),
    # streaming 1x1 kernel (k_conv1.h)
    (BF16, 2, 5, 77, 64, 128, 1, 2, 0),    # projection shortcut, stride 2, odd width
    (BF16, 1, 3, 40, 8, 64, 1, 1, 0),      # 8 input channels (one 16-channel k-step)
    (BF16, 1, 4, 33, 128, 128, 1, 1, 6),   # residual + relu, single-buffered variant
    (BF16, 2, 3, 50, 64, 64, 1, 2, 4),
    # fp16 (RD_F16): the same kernels on v_mfma_f32_32x32x16_f16 -- persistent 3x3 (plain epilogue), stride 2, 72-channel input,
    # generic tap kernel for the 1x1 shapes
    (F16, 2, 9, 70, 64, 128, 3, 1, 6), (F16, 1, 6, 127, 128, 64, 3, 1, 5), (F16, 1, 4, 131, 128, 128, 3, 2, 4),
    (F16, 1, 2, 20, 72, 128, 3, 1, 4), (F16, 2, 5, 77, 64, 128, 1, 2, 0), (F16, 1, 4, 33, 128, 128, 1, 1, 6),
]


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_bn_act(be, case):
    dt, B, H, W, cin, cout, k, s, fl = case
    run_conv(be, dt, B, H, W, cin, cout, k, s, fl, cs_in=80 if cin == 72 else None)


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(2, 9, 70, 128, 8), (1, 4, 130, 128, 1), (1, 8, 62, 72, 7), (2, 9, 70, 128, 8, F16), (1, 8, 62, 72, 1, F16)])
def test_conv_fused_with_head_out(be, case):
    """rd_conv2d_bn_act_head_out == rd_conv2d_bn_act (bf16, ReLU) followed by rd_head_out on its output: the same bf16
    activations feed the same hi + lo weight MFMAs, so only the fp32 summation order differs."""
    B, H, W, cin, nout = case[:5]
    BF16 = case[5] if len(case) > 5 else R.RD_BF16        # (the element type under test; the body below reads "BF16")
    rng = np.random.default_rng(5)
    x = h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), BF16)
    w = h16_round((rng.standard_normal((128, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), BF16)
    sc, sh = rng.uniform(0.5, 1.5, 128).astype(np.float32), rng.standard_normal(128).astype(np.float32)
    hw = (rng.standard_normal((nout, 128)) / np.sqrt(128)).astype(np.float32)
    hb = rng.standard_normal(nout).astype(np.float32)
    cs = -(-cin // 16) * 16
    L = be.lib
    xin = be.up(to_nhwc(x, BF16, cstride=cs))
    wp, dsc, dsh = be.up(L.pack_conv_weight(w, BF16)), be.up(sc), be.up(sh)
    dhw, dhb, dhp = be.up(hw), be.up(hb), be.up(L.pack_head_weight(hw, dtype=BF16))
    N, off = H * W + 37, 21                                          # a level's slice of a longer flat tensor
    y = be.empty(B * H * W * 128 * 2)
    o1, o2 = be.empty(B * N * nout * 4), be.empty(B * N * nout * 4)
    L.call("rd_conv2d_bn_act", be.ptr(xin), cs, 0, be.ptr(wp), be.ptr(dsc), be.ptr(dsh), None, 0, 0, be.ptr(y), 128, 0, B, H, W,
           cin, 128, 3, 3, 1, R.RD_RELU_POST, BF16, be.stream)
    L.call("rd_head_out", be.ptr(y), 128, 0, be.ptr(dhw), be.ptr(dhb), be.ptr(o1), N * nout, off, B, H, W, 128, nout, BF16, be.stream)
    a = be.down(o1, np.float32, (B, N, nout))
    assert np.abs(a[:, off:off + H * W]).max() > 0.5
    # (the emulator build instantiates the fp16 persistent kernel for the production = folded-scale forms only, k_conv3.h
    #  conv3_has_form: the un-folded fused launch is checked on the GPU, and in bf16 everywhere)
    unfolded = not (BF16 == R.RD_F16 and be.name == "emu")
    if unfolded:
        L.call("rd_conv2d_bn_act_head_out", be.ptr(xin), cs, 0, be.ptr(wp), be.ptr(dsc), be.ptr(dsh), B, H, W, cin, R.RD_RELU_POST,
               be.ptr(dhp), be.ptr(dhb), be.ptr(o2), N * nout, off, nout, BF16, be.stream)
        b = be.down(o2, np.float32, (B, N, nout))
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), np.abs(a - b).max()
        assert (b[:, :off] == 0).all() and (b[:, off + H * W:] == 0).all()   # nothing outside the level's slice is touched
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy() * sc[None, :, None, None] + sh[None, :, None, None]
    act = h16_round(np.maximum(ref, 0).astype(np.float32), BF16)
    want = np.einsum('oc,bchw->bhwo', hw, act).reshape(B, H * W, nout) + hb
    if unfolded:
        assert np.abs(b[:, off:off + H * W] - want).max() < 2 * _ulp(BF16) * max(1.0, np.abs(want).max())
    # the production form: scale folded into the 3x3 weights (RD_SCALE_FOLDED), shift through the accumulators
    o3 = be.empty(B * N * nout * 4)
    wpf = be.up(L.pack_conv3x3_ex(w, 1, cs, fold_scale=sc, dtype=BF16))
    L.call("rd_conv2d_bn_act_head_out", be.ptr(xin), cs, 0, be.ptr(wpf), None, be.ptr(dsh), B, H, W, cin,
           R.RD_RELU_POST | R.RD_SCALE_FOLDED, be.ptr(dhp), be.ptr(dhb), be.ptr(o3), N * nout, off, nout, BF16, be.stream)
    c = be.down(o3, np.float32, (B, N, nout))
    assert np.abs(c[:, off:off + H * W] - want).max() < 3 * _ulp(BF16) * max(1.0, np.abs(want).max())
    assert (c[:, :off] == 0).all() and (c[:, off + H * W:] == 0).all()
    if cin % 32 == 0:
        # ... and its 16 x 16 x 32 MFMA form (RD_MFMA16): every product is exact in fp32 and each accumulator adds the same 32-channel
        # groups in the same order, so the tower activations are the SAME numbers; the output conv sums them in another order
        assert L.raw("rd_conv3x3_mfma16_ok")(cin, 128, 1, W, 1) == 1 and L.raw("rd_conv3x3_mfma16_ok")(72, 128, 1, W, 1) == 0
        o4 = be.empty(B * N * nout * 4)
        wp16 = be.up(L.pack_conv3x3_m16(w, sc, dtype=BF16))
        dhp16 = be.up(L.pack_head_weight(hw, dtype=BF16, m16=True))      # (its own head-weight image: 16-row fragments, pack_head_frag16)
        L.call("rd_conv2d_bn_act_head_out", be.ptr(xin), cs, 0, be.ptr(wp16), None, be.ptr(dsh), B, H, W, cin,
               R.RD_RELU_POST | R.RD_SCALE_FOLDED | R.RD_MFMA16, be.ptr(dhp16), be.ptr(dhb), be.ptr(o4), N * nout, off, nout, BF16, be.stream)
        d = be.down(o4, np.float32, (B, N, nout))
        assert np.abs(d[:, off:off + H * W] - want).max() < 3 * _ulp(BF16) * max(1.0, np.abs(want).max())
        assert np.abs(d - c).max() <= 2e-5 * max(1.0, np.abs(c).max()), np.abs(d - c).max()
        assert (d[:, :off] == 0).all() and (d[:, off + H * W:] == 0).all()
    buf = be.ptr(be.empty(1 << 16))
    # RD_MFMA16 where the library has no such form (72 input channels): refused, not mis-launched
    assert L.raw("rd_conv2d_bn_act_head_out")(buf, 80, 0, buf, None, buf, 1, 4, 8, 72, 4 | 8 | 16, buf, buf, buf, 100, 0, 8, BF16, be.stream) == R.RD_ESHAPE
    assert L.raw("rd_conv2d_bn_act_head_out")(buf, 128, 0, buf, buf, buf, 1, 4, 8, 128, 4, buf, buf, buf, 100, 0, 9, BF16, be.stream) == R.RD_ESHAPE
    assert L.raw("rd_conv2d_bn_act_head_out")(buf, 128, 0, buf, buf, buf, 1, 4, 8, 128, 6, buf, buf, buf, 100, 0, 8, BF16, be.stream) == R.RD_EINVAL
    assert L.raw("rd_conv2d_bn_act_head_out")(buf, 128, 0, buf, buf, buf, 1, 4, 8, 128, 4, buf, buf, buf, 100, 0, 8, F32, be.stream) == R.RD_EINVAL


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(2, 9, 70, 128, True, BF16), (1, 8, 64, 72, False, BF16), (3, 17, 40, 64, True, BF16), (2, 9, 70, 128, True, F16),
                                  (1, 8, 33, 72, False, F16)])
def test_conv_pair_equals_two_launches(be, case):
    if be.name == "emu" and case[:3] in ((3, 17, 40), (2, 9, 70)) and case[5] == F16:
        pytest.skip("CPU tier: the fp16 forms of the larger shapes run on the GPU only (emulator time)")
    """rd_conv3x3_bn_act_pair / rd_conv2d_bn_act_head_out_pair (the cls and reg tower conv of a head level as ONE launch,
    head/builder.py:221-261) == the two single launches, bit for bit: same tiles, same MFMA order, only the tile list is shared.
    Shapes: tile lists that cross from problem 0 into problem 1 inside a workgroup, workgroups that START in problem 1 (fewer
    tiles than resident slots), a shared input (layer 0 of the towers) and separate inputs (layers 1-3)."""
    B, H, W, cin, shared_x, dt = case
    rng = np.random.default_rng(11)
    cs = -(-cin // 16) * 16
    L = be.lib
    xs = [h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), dt) for _ in range(2)]
    if shared_x:
        xs[1] = xs[0]
    ws = [h16_round((rng.standard_normal((128, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dt) for _ in range(2)]
    scs = [rng.uniform(0.5, 1.5, 128).astype(np.float32) for _ in range(2)]
    shs = [rng.standard_normal(128).astype(np.float32) for _ in range(2)]
    dx = [be.up(to_nhwc(x, dt, cstride=cs)) for x in xs]
    if shared_x:
        dx[1] = dx[0]
    dw = [be.up(L.pack_conv3x3_ex(w, 1, cs, fold_scale=sc, dtype=dt)) for w, sc in zip(ws, scs)]
    dsh = [be.up(sh) for sh in shs]
    fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED
    nb = B * H * W * 128 * 2
    ys, yp = [be.empty(nb) for _ in range(2)], [be.empty(nb) for _ in range(2)]
    for g in range(2):
        L.call("rd_conv3x3_bn_act_ex", be.ptr(dx[g]), cs, 0, be.ptr(dw[g]), None, be.ptr(dsh[g]), None, 0, 0, None, 0, 0, 0, None,
               be.ptr(ys[g]), 128, 0, B, H, W, cin, 128, 1, fl, dt, be.stream)
    L.call("rd_conv3x3_bn_act_pair", be.ptr(dx[0]), 0, be.ptr(dw[0]), be.ptr(dsh[0]), be.ptr(yp[0]), 0,
           be.ptr(dx[1]), 0, be.ptr(dw[1]), be.ptr(dsh[1]), be.ptr(yp[1]), 0, cs, 128, B, H, W, cin, fl, dt, be.stream)
    for g in range(2):
        a, b = be.down(ys[g], np.uint16, (B, H, W, 128)), be.down(yp[g], np.uint16, (B, H, W, 128))
        assert np.abs(from_nhwc(a, dt, 128)).max() > 0.5
        assert np.array_equal(a, b), (g, int((a != b).sum()))
    # torch fp32 reference of problem 1 (the single launch is checked against it elsewhere; this pins the pair on its own)
    ref = F.conv2d(torch.from_numpy(xs[1]), torch.from_numpy(ws[1]), padding=1).numpy() * scs[1][None, :, None, None] + shs[1][None, :, None, None]
    ref = np.maximum(ref, 0)
    got = from_nhwc(be.down(yp[1], np.uint16, (B, H, W, 128)), dt, 128)
    assert np.abs(got - ref).max() <= 3 * _ulp(dt) * max(1.0, float(np.abs(ref).max()))   # (folded scale: w*s is rounded, not w)
    # the towers' last convs with their 1x1 output convs: logits (1 output) and deltas (8 outputs) of one level
    nouts = (1, 8)
    hws = [(rng.standard_normal((n, 128)) / np.sqrt(128)).astype(np.float32) for n in nouts]
    hbs = [rng.standard_normal(n).astype(np.float32) for n in nouts]
    dhp = [be.up(L.pack_head_weight(hw, dtype=dt)) for hw in hws]
    dhb = [be.up(hb) for hb in hbs]
    N, off = H * W + 29, 13
    os_, op = [be.empty(B * N * n * 4) for n in nouts], [be.empty(B * N * n * 4) for n in nouts]
    for g in range(2):
        L.call("rd_conv2d_bn_act_head_out", be.ptr(dx[g]), cs, 0, be.ptr(dw[g]), None, be.ptr(dsh[g]), B, H, W, cin, fl,
               be.ptr(dhp[g]), be.ptr(dhb[g]), be.ptr(os_[g]), N * nouts[g], off, nouts[g], dt, be.stream)
    L.call("rd_conv2d_bn_act_head_out_pair",
           be.ptr(dx[0]), 0, be.ptr(dw[0]), be.ptr(dsh[0]), be.ptr(dhp[0]), be.ptr(dhb[0]), be.ptr(op[0]), N * nouts[0], nouts[0],
           be.ptr(dx[1]), 0, be.ptr(dw[1]), be.ptr(dsh[1]), be.ptr(dhp[1]), be.ptr(dhb[1]), be.ptr(op[1]), N * nouts[1], nouts[1],
           cs, off, B, H, W, cin, fl, dt, be.stream)
    for g in range(2):
        a, b = be.down(os_[g], np.float32, (B, N, nouts[g])), be.down(op[g], np.float32, (B, N, nouts[g]))
        assert np.abs(a[:, off:off + H * W]).max() > 0.3
        assert np.array_equal(a, b), (g, float(np.abs(a - b).max()))
        assert (b[:, :off] == 0).all() and (b[:, off + H * W:] == 0).all()
    buf = be.ptr(be.empty(1 << 16))
    f = L.raw("rd_conv3x3_bn_act_pair")
    assert f(buf, 0, buf, buf, buf, 0, buf, 0, buf, buf, buf, 0, 128, 128, 1, 4, 8, 128, R.RD_RELU_POST, dt, be.stream) == R.RD_EINVAL      # not folded
    assert f(buf, 0, buf, buf, buf, 0, buf, 0, buf, None, buf, 0, 128, 128, 1, 4, 8, 128, fl, dt, be.stream) == R.RD_EINVAL               # null shift
    assert f(buf, 0, buf, buf, buf, 64, buf, 0, buf, buf, buf, 0, 128, 128, 1, 4, 8, 128, fl, dt, be.stream) == R.RD_ESHAPE               # y channels
    assert f(buf, 0, buf, buf, buf, 0, buf, 0, buf, buf, buf, 0, 128, 128, 1, 4, 8, 128, fl, F32, be.stream) == R.RD_EINVAL


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(F32, 1, 3, 10, 128, 64, (3, 8), 4, 2), (BF16, 1, 2, 20, 128, 128, (3, 8), 4, 2),
                                  (F32, 2, 2, 13, 64, 64, (3, 4), 2, 1), (BF16, 1, 3, 24, 128, 64, (3, 4), 2, 1),
                                  (F16, 1, 2, 20, 128, 128, (3, 8), 4, 2), (F16, 1, 3, 24, 128, 64, (3, 4), 2, 1)])
def test_deconv2d_bn_act(be, case):
    dt, B, H, W, cin, cout, k, s, pw = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k[0], k[1])) / np.sqrt(cin * k[0] * k[1] / s)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    Wout = (W - 1) * s - 2 * pw + k[1]
    res = rng.standard_normal((B, cout, H, Wout)).astype(np.float32)
    x, w, res = h16_round(x, dt), h16_round(w, dt), h16_round(res, dt)
    L = be.lib
    xin, rin = be.up(to_nhwc(x, dt)), be.up(to_nhwc(res, dt))
    y = be.empty(B * H * Wout * cout * 4)
    dsc, dsh = be.up(sc), be.up(sh)
    for ph in range(s):
        wp = be.up(L.pack_deconv_weight(w, s, pw, ph, dt))
        L.call("rd_deconv2d_bn_act", be.ptr(xin), cin, 0, be.ptr(wp), be.ptr(dsc), be.ptr(dsh), be.ptr(rin), cout, 0,
               be.ptr(y), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, ph, R.RD_RELU_PRE | R.RD_ADD, dt, be.stream)
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), stride=(1, s), padding=(1, pw)).numpy()
    ref = np.maximum(ref * sc[None, :, None, None] + sh[None, :, None, None], 0) + res
    got = from_nhwc(be.down(y, np.uint16 if dt in H16 else np.float32, (B, H, Wout, cout)), dt, cout)
    assert np.abs(got - ref).max() <= _tol(dt, ref)
    if dt in H16:   # the production form: BatchNorm scale folded into the weights, shift through the accumulators
        y2 = be.empty(B * H * Wout * cout * 4)
        for ph in range(s):
            wp = be.up(L.pack_deconv_weight(w, s, pw, ph, dt, fold_scale=sc))
            L.call("rd_deconv2d_bn_act", be.ptr(xin), cin, 0, be.ptr(wp), None, be.ptr(dsh), be.ptr(rin), cout, 0, be.ptr(y2), cout, 0,
                   B, H, W, cin, cout, k[0], k[1], s, pw, ph, R.RD_RELU_PRE | R.RD_ADD | R.RD_SCALE_FOLDED, dt, be.stream)
        got2 = from_nhwc(be.down(y2, np.uint16, (B, H, Wout, cout)), dt, cout)
        assert np.abs(got2 - ref).max() <= 1.5 * _tol(dt, ref)


@pytest.mark.parametrize("be", WITH_LATE_DMA, indirect=True)
@pytest.mark.parametrize("case", [(BF16, 2, 9, 70, 128, 128, (3, 8), 4, 2), (BF16, 3, 17, 40, 128, 64, (3, 4), 2, 1), (BF16, 1, 20, 45, 64, 64, (3, 4), 2, 1),
                                  (F16, 2, 9, 70, 128, 64, (3, 8), 4, 2), (F16, 1, 11, 33, 64, 64, (3, 4), 2, 1)])
def test_deconv2d_all_phases_in_one_launch(be, case):
    """rd_deconv2d_bn_act_all (every output phase of a transposed conv in ONE launch: the tile list is (tile, phase), the tap-set
    side and the weight image change per list entry) == the stride_w per-phase launches of rd_deconv2d_bn_act, bit for bit, and
    within one output rounding of torch's conv_transpose2d (mxnext/simple.py:545-580, dla_backbone.py:117-127).  Shapes: several
    tiles per workgroup so that phases and tiles alternate inside a list, partial column / row tiles, batch > 1."""
    dt, B, H, W, cin, cout, k, s, pw = case
    rng = np.random.default_rng(21)
    x = h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), dt)
    w = (rng.standard_normal((cin, cout, k[0], k[1])) / np.sqrt(cin * k[0] * k[1] / s)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    Wout = (W - 1) * s - 2 * pw + k[1]
    res = h16_round(rng.standard_normal((B, cout, H, Wout)).astype(np.float32), dt)
    L = be.lib
    assert L.raw("rd_deconv2d_all_phases_ok")(k[0], k[1], s, pw, cout, dt) == 1
    assert L.raw("rd_deconv2d_all_phases_ok")(3, 3, 2, 1, cout, dt) == 0          # Wout != 2 Win
    assert L.raw("rd_deconv2d_all_phases_ok")(k[0], k[1], s, pw, cout, F32) == 0
    xin, rin, dsh = be.up(to_nhwc(x, dt)), be.up(to_nhwc(res, dt)), be.up(sh)
    fl = R.RD_RELU_PRE | R.RD_ADD | R.RD_SCALE_FOLDED
    imgs = [L.pack_deconv_weight(w, s, pw, ph, dt, fold_scale=sc) for ph in range(s)]
    y1, y2 = be.empty(B * H * Wout * cout * 2), be.empty(B * H * Wout * cout * 2)
    for ph in range(s):
        L.call("rd_deconv2d_bn_act", be.ptr(xin), cin, 0, be.ptr(be.up(imgs[ph])), None, be.ptr(dsh), be.ptr(rin), cout, 0, be.ptr(y1), cout, 0,
               B, H, W, cin, cout, k[0], k[1], s, pw, ph, fl, dt, be.stream)
    L.call("rd_deconv2d_bn_act_all", be.ptr(xin), cin, 0, be.ptr(be.up(np.concatenate(imgs))), len(imgs[0]), be.ptr(dsh), be.ptr(rin), cout, 0,
           be.ptr(y2), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, fl, dt, be.stream)
    a, b = be.down(y1, np.uint16, (B, H, Wout, cout)), be.down(y2, np.uint16, (B, H, Wout, cout))
    assert np.array_equal(a, b), int((a != b).sum())
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), stride=(1, s), padding=(1, pw)).numpy()
    ref = np.maximum(ref * sc[None, :, None, None] + sh[None, :, None, None], 0) + res
    assert np.abs(from_nhwc(b, dt, cout) - ref).max() <= 1.5 * _tol(dt, ref)
    buf = be.ptr(be.empty(1 << 16))
    f = L.raw("rd_deconv2d_bn_act_all")
    assert f(buf, 128, 0, buf, 1 << 20, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 8, 4, 2, R.RD_RELU_PRE | R.RD_ADD, dt, be.stream) == R.RD_EINVAL   # not folded
    assert f(buf, 128, 0, buf, 1 << 20, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 3, 2, 1, fl, dt, be.stream) == R.RD_ESHAPE                       # not 3 x 2 phases
    assert f(buf, 128, 0, buf, 16, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 8, 4, 2, fl, dt, be.stream) == R.RD_EINVAL                             # phase images overlap


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(BF16, 2, 9, 70, 128, True), (BF16, 3, 17, 40, 64, True), (F16, 1, 11, 33, 128, True), (BF16, 1, 8, 64, 128, False)])
def test_deconv2d_phase_pairs_equal_all_phases(be, case):
    """rd_deconv2d_bn_act_pairs (output phases 2p | 2p+1 of the k(3,8) stride-4 transposed conv as ONE 128-channel problem on the
    cout-128 form of the persistent kernel, dla_backbone.py:117-127 'agg1') == rd_deconv2d_bn_act_all bit for bit: the K order of
    every output element is the same.  With and without the residual; error codes for layers / tensors without the form."""
    dt, B, H, W, cin, with_res = case
    cout, k, s, pw = 64, (3, 8), 4, 2
    rng = np.random.default_rng(33)
    x = h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), dt)
    w = (rng.standard_normal((cin, cout, k[0], k[1])) / np.sqrt(cin * k[0] * k[1] / s)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32) * np.where(rng.random(cout) < 0.2, -1, 1).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    Wout = s * W
    res = h16_round(rng.standard_normal((B, cout, H, Wout)).astype(np.float32), dt)
    L = be.lib
    ok = L.raw("rd_deconv2d_phase_pairs_ok")
    assert ok(3, 8, 4, 2, 64, dt) == 1
    assert ok(3, 4, 2, 1, 64, dt) == 0          # stride 2: the two phases read different column pairs
    assert ok(3, 8, 4, 2, 128, dt) == 0         # a pair would have 256 output channels
    assert ok(3, 8, 4, 2, 64, F32) == 0
    xin, rin = be.up(to_nhwc(x, dt)), be.up(to_nhwc(res, dt))
    fl = (R.RD_RELU_PRE | R.RD_ADD | R.RD_SCALE_FOLDED) if with_res else (R.RD_RELU_POST | R.RD_SCALE_FOLDED)
    rp = be.ptr(rin) if with_res else None
    imgs = [L.pack_deconv_weight(w, s, pw, ph, dt, fold_scale=sc) for ph in range(s)]
    pairs = [L.pack_deconv_phase_pair(imgs[p], imgs[p + 1], cin, dt) for p in (0, 2)]
    y1, y2 = be.empty(B * H * Wout * cout * 2), be.empty(B * H * Wout * cout * 2)
    L.call("rd_deconv2d_bn_act_all", be.ptr(xin), cin, 0, be.ptr(be.up(np.concatenate(imgs))), len(imgs[0]), be.ptr(be.up(sh)), rp, cout, 0,
           be.ptr(y1), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, fl, dt, be.stream)
    L.call("rd_deconv2d_bn_act_pairs", be.ptr(xin), cin, 0, be.ptr(be.up(np.concatenate(pairs))), len(pairs[0]), be.ptr(be.up(np.concatenate([sh, sh]))),
           rp, cout, 0, be.ptr(y2), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, fl, dt, be.stream)
    a, b = be.down(y1, np.uint16, (B, H, Wout, cout)), be.down(y2, np.uint16, (B, H, Wout, cout))
    assert np.array_equal(a, b), int((a != b).sum())
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), stride=(1, s), padding=(1, pw)).numpy() * sc[None, :, None, None] + sh[None, :, None, None]
    ref = (np.maximum(ref, 0) + res) if with_res else np.maximum(ref, 0)
    assert np.abs(from_nhwc(b, dt, cout) - ref).max() <= 1.5 * _tol(dt, ref)
    buf = be.ptr(be.empty(1 << 16))
    f = L.raw("rd_deconv2d_bn_act_pairs")
    assert f(buf, 128, 0, buf, 1 << 20, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 4, 2, 1, fl, dt, be.stream) == R.RD_ESHAPE      # stride 2
    assert f(buf, 128, 0, buf, 1 << 20, buf, buf, 64, 0, buf, 80, 0, 1, 4, 8, 128, 64, 3, 8, 4, 2, fl, dt, be.stream) == R.RD_ESHAPE      # output not dense
    assert f(buf, 128, 0, buf, 1 << 20, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 8, 4, 2, fl & ~R.RD_SCALE_FOLDED, dt, be.stream) == R.RD_EINVAL
    assert f(buf, 128, 0, buf, 4096, buf, buf, 64, 0, buf, 64, 0, 1, 4, 8, 128, 64, 3, 8, 4, 2, fl, dt, be.stream) == R.RD_EINVAL          # pair images overlap


@pytest.mark.parametrize("be", WITH_LATE_DMA, indirect=True)
@pytest.mark.parametrize("case", [(F32, 1, 3, 40), (F32, 1, 9, 33), (BF16, 2, 10, 40), (F16, 2, 10, 40), (BF16, 8, 17, 70), (BF16, 4, 8, 33)])
def test_meta_kernel_unit(be, case):
    """Fused Meta-Kernel unit vs the un-fused restatement of meta_kernel.py:166-240 + dla_backbone.py:92-97.  (8 images x 3 column
    tiles: the strips divide by 8 -> the XCD-aware tile order, MetaArgs::r0 = 8, several tiles per workgroup on the emulator; 4 x 8 x
    33: one row tile -- the decode's divide-by-one case.)"""
    dt, B, H, W = case
    rng = np.random.default_rng(0)
    P = synth.make_weights(seed=18, width=W)
    data = rng.standard_normal((B, 64, H, W)).astype(np.float32)
    coord = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    data = h16_round(data, dt)
    name = 'res1_unit2'
    ref = G.meta_kernel_unit(G.T(data), G.T(coord), P, name).numpy()
    s1, t1 = bn_affine(P, name + "point_wise_mlp_bn1", G.EPS)
    s2, t2 = bn_affine(P, name + "aggregation_bn1", G.EPS)
    pre = name + "_%d" % W
    L = be.lib
    pk = L.pack_meta(P[pre + "_mlp0_weight"].reshape(32, 3), P[pre + "_mlp0_bias"], P[pre + "_mlp1_weight"].reshape(64, 32),
                     P[pre + "_mlp1_bias"], s1, t1, P[name + "aggregation_conv1_weight"].reshape(64, 576), s2, t2, dt)
    x, c, pkd = be.up(to_nhwc(data, dt)), be.up(coord), be.up(pk)
    y = be.empty(B * H * W * 64 * 4)
    L.call("rd_meta_kernel_fwd", be.ptr(x), 64, 0, be.ptr(c), be.ptr(pkd), be.ptr(y), 64, 0, B, H, W, dt, be.stream)
    got = from_nhwc(be.down(y, np.uint16 if dt in H16 else np.float32, (B, H, W, 64)), dt, 64)
    # fp16: the same model with 2^-12 in place of 2^-9.  bf16 error model: the hidden vector, the 576 products and both weight sets are rounded to bf16 (2^-9 relative, rms
    # 2^-9/sqrt(3) each): four independent roundings per term of the 576-term sum -> relative rms error 2^-9*sqrt(4/3) of the
    # pre-activation spread; the output is rounded once more (2^-9 of its magnitude).  Allow 6 sigma over the 1e5 outputs.
    if dt == F32:
        tol = 1e-4
    else:
        u = 2.0 ** -9 if dt == BF16 else 2.0 ** -12
        rel = u * np.sqrt(4.0 / 3.0)
        err = got - ref
        print("meta 16-bit: rms err / std %.5f (model %.5f), max err / std %.4f" % (err.std() / ref.std(), rel, np.abs(err).max() / ref.std()))
        assert err.std() < 2.0 * rel * ref.std()
        tol = 6 * 2.0 * rel * float(ref.std()) + u * float(np.abs(ref).max()) + 1e-4
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_head_out_and_layout(be, dt):
    rng = np.random.default_rng(2)
    B, H, W, C = 2, 3, 37, 128
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    x = h16_round(x, dt)
    L = be.lib
    src = be.up(x)
    xin = be.empty(B * H * W * C * 4)
    L.call("rd_nchw_to_nhwc", be.ptr(src), be.ptr(xin), B, C, H, W, C, 0, 0, dt, be.stream)
    back = be.empty(B * H * W * C * 4)
    L.call("rd_nhwc_to_nchw", be.ptr(xin), be.ptr(back), B, C, H, W, C, 0, dt, be.stream)
    assert np.array_equal(be.down(back, np.float32, (B, C, H, W)), x)
    # few channels into a wider zero-padded buffer (the 8-channel range image next to the 64 agg channels: lower.py concat)
    x8 = x[:, :8].copy()
    cs, co, pad = 80, 16, 8 if dt in H16 else 0
    esz = 2 if dt in H16 else 4
    wide = be.up(np.full(B * H * W * cs * esz, 0x5A, np.uint8))
    L.call("rd_nchw_to_nhwc", be.ptr(be.up(x8)), be.ptr(wide), B, 8, H, W, cs, co, pad, dt, be.stream)
    got = be.down(wide, np.uint16 if dt in H16 else np.float32, (B, H, W, cs))
    ref8 = np.transpose(x8, (0, 2, 3, 1))
    if dt in H16:
        assert np.array_equal(got[..., co:co + 8], to_nhwc(x8, dt))
        assert not got[..., co + 8:co + 16].any() and (got[..., :co] == 0x5A5A).all() and (got[..., co + 16:] == 0x5A5A).all()
    else:
        assert np.array_equal(got[..., co:co + 8], ref8)
    N = H * W + 50
    for nout in (1, 7, 8):
        w = (rng.standard_normal((nout, C)) * 0.1).astype(np.float32)
        b = rng.standard_normal(nout).astype(np.float32)
        out = be.empty(B * N * nout * 4)
        L.call("rd_head_out", be.ptr(xin), C, 0, be.ptr(be.up(w)), be.ptr(be.up(b)), be.ptr(out), N * nout, 50, B, H, W, C,
               nout, dt, be.stream)
        got = be.down(out, np.float32, (B, N, nout))[:, 50:]
        ref = np.einsum("bchw,oc->bhwo", x, w).reshape(B, H * W, nout) + b
        assert np.abs(got - ref).max() < 1e-4


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_sorted_foreground(be):
    rng = np.random.default_rng(0)
    B, N, k, D = 2, 5000, 1200, 8
    logit = rng.standard_normal((B, N)).astype(np.float32)
    logit[:, ::7] = logit[:, 3:4]  # exact ties among positive scores
    mask = (rng.uniform(size=(B, N)) > 0.4).astype(np.float32)  # > k masked-out zeros: ties at 0 as well
    delta = rng.standard_normal((B, N, D)).astype(np.float32)
    pc = rng.standard_normal((B, N, 3)).astype(np.float32)
    score = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    L = be.lib
    rs, rd, rp, ri = O.get_sorted_foreground(score, delta, pc, mask, k)
    # workspace for one problem (batch elements sorted one after the other) and for B (sorted side by side)
    for mult in (1, B):
        nb = L.raw("rd_sorted_foreground_workspace_bytes")(N, k) * mult
        ws = be.empty(nb)
        o_s, o_d, o_p, o_i = be.empty(B * k * 4), be.empty(B * k * D * 4), be.empty(B * k * 12), be.empty(B * k * 4)
        L.call("rd_sorted_foreground", be.ptr(be.up(score)), be.ptr(be.up(delta)), be.ptr(be.up(pc)), be.ptr(be.up(mask)), B, N,
               k, D, 0, be.ptr(o_s), be.ptr(o_d), be.ptr(o_p), be.ptr(o_i), be.ptr(ws), nb, be.stream)
        assert np.array_equal(be.down(o_i, np.int32, (B, k)), ri)           # bit-exact index work, incl. tie rule
        assert np.array_equal(be.down(o_s, np.float32, (B, k)), rs)
        assert np.array_equal(be.down(o_d, np.float32, (B, k, D)), rd)
        assert np.array_equal(be.down(o_p, np.float32, (B, k, 3)), rp)
    # N < k mirrors the reference's assert (get_sorted_foreground.py:65)
    with pytest.raises(R.RangeDetError) as e:
        L.call("rd_sorted_foreground", be.ptr(o_s), be.ptr(o_d), be.ptr(o_p), None, 1, 10, 20, D, 0, be.ptr(o_s), be.ptr(o_d),
               be.ptr(o_p), None, be.ptr(ws), nb, be.stream)
    assert e.value.code == R.RD_ESHAPE


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", ["all_equal", "top_ties", "half", "narrow"])
def test_sorted_foreground_preselect_edges(be, case):
    """The top-k pre-selection (12-bit threshold bin, ordered compaction, sort of the candidates only) against the full-sort
    oracle on the distributions that stress it: every key in one bin, ties across the cut, k = N / 2 exactly, scores packed
    into a few bins."""
    rng = np.random.default_rng(11)
    B, N, D = 2, 6000, 8
    k = N // 2 if case == "half" else 1500
    if case == "all_equal":
        score = np.full((B, N), 0.7310586, np.float32)
    elif case == "top_ties":
        score = rng.uniform(0.0, 0.9, (B, N)).astype(np.float32)
        score[:, rng.permutation(N)[:2000]] = 0.95           # 2000 equal best scores: the cut at k = 1500 falls inside the tie
    elif case == "narrow":
        score = (0.5 + rng.integers(0, 4, (B, N)) * 2.0 ** -12).astype(np.float32)   # four distinct values, one 12-bit bin
    else:
        score = rng.uniform(0.0, 1.0, (B, N)).astype(np.float32)
    delta = rng.standard_normal((B, N, D)).astype(np.float32)
    pc = rng.standard_normal((B, N, 3)).astype(np.float32)
    L = be.lib
    mask = np.ones((B, N), np.float32)
    rs, rd, rp, ri = O.get_sorted_foreground(score, delta, pc, mask, k)
    nb = L.raw("rd_sorted_foreground_workspace_bytes")(N, k) * B
    ws = be.empty(nb)
    o_s, o_d, o_p, o_i = be.empty(B * k * 4), be.empty(B * k * D * 4), be.empty(B * k * 12), be.empty(B * k * 4)
    L.call("rd_sorted_foreground", be.ptr(be.up(score)), be.ptr(be.up(delta)), be.ptr(be.up(pc)), be.ptr(be.up(mask)), B, N, k, D, 0,
           be.ptr(o_s), be.ptr(o_d), be.ptr(o_p), be.ptr(o_i), be.ptr(ws), nb, be.stream)
    assert np.array_equal(be.down(o_i, np.int32, (B, k)), ri)
    assert np.array_equal(be.down(o_s, np.float32, (B, k)), rs)
    assert np.array_equal(be.down(o_d, np.float32, (B, k, D)), rd)


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_decode3d_bbox(be):
    rng = np.random.default_rng(3)
    B, N = 2, 3000
    d = rng.normal(0, 0.7, (B, N, 8)).astype(np.float32)
    pc = rng.uniform(-70, 70, (B, N, 3)).astype(np.float32)
    pc[0, 0] = 0          # point at the origin: atan2(0,0)
    d[0, 1, :2] = 0       # zero offsets
    d[0, 2, 2:4] = 3.0    # large log sizes
    L = be.lib
    out = be.empty(B * N * 40)
    L.call("rd_decode3d_bbox", be.ptr(be.up(d)), be.ptr(be.up(pc)), be.ptr(out), B, N, 8, 0, be.stream)
    ref = O.decode3d(d, pc)
    # device libm (atan2f/sinf/cosf/expf) differs from glibc by ulps: 1e-4 is the north-star box tolerance
    assert np.abs(be.down(out, np.float32, (B, N, 10)) - ref).max() < 1e-4
    d7 = rng.normal(0, 0.5, (B, N, 7)).astype(np.float32)
    L.call("rd_decode3d_bbox", be.ptr(be.up(d7)), be.ptr(be.up(pc)), be.ptr(out), B, N, 7, 1, be.stream)
    assert np.abs(be.down(out, np.float32, (B, N, 10)) - O.decode3d(d7, pc, True)).max() < 1e-4
    with pytest.raises(R.RangeDetError):
        L.call("rd_decode3d_bbox", be.ptr(out), be.ptr(out), be.ptr(out), B, N, 9, 0, be.stream)


def _wnms(be, d, thr, vote, is3d, order=None, cap_extra=5, tie=R.RD_TIE_STABLE, hash_scale=100):
    L = be.lib
    K = d.shape[0]
    cap = K + cap_extra
    dbuf = np.zeros((cap, 12), np.float32)
    dbuf[:K] = d
    ob = None
    if order is not None:
        ob = np.zeros(cap, np.int32)
        ob[:K] = order
    nb = L.raw("rd_wnms_workspace_bytes")(cap)
    ws, outd, keep, nk = be.empty(nb), be.empty(cap * 48), be.empty(cap * 4), be.empty(16)
    L.call("rd_wnms_4c", be.ptr(be.up(dbuf)), cap, be.ptr(be.up(np.array([K], np.int32))), be.ptr(be.up(ob)) if ob is not None else None,
           tie, thr, vote, int(is3d), hash_scale, be.ptr(outd), be.ptr(keep), be.ptr(nk), be.ptr(ws), nb, be.stream)
    M = int(be.down(nk, np.int32, (1,))[0])
    return be.down(outd, np.float32, (cap, 12))[:M], be.down(keep, np.int32, (cap,))[:M]


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(12, 6, None, 0), (20, 9, 33, 0), (10, 5, None, 1), (1, 1, None, 0), (1, 2, None, 0), (1, 3, None, 0)])
def test_wnms_vs_oracle(be, case):
    """keep indices bit-exact and merged rows bit-equal, with the reference ordering (std::sort on the host, ties
    included) and with the on-device ordering (score desc, index asc)."""
    no, rep, quant, is3d = case
    d = synth.cluster_dets(no, rep, seed=no + rep, quant=quant)
    order = be.lib.wnms_order_host(d)
    assert np.array_equal(order, O.wnms_order(d))
    rows, keep = _wnms(be, d, 0.1, 0.5, is3d, order)
    flat, rk = O.wnms_4c(d, 0.1, 0.5, bool(is3d), 100)
    assert keep.tolist() == rk
    assert np.array_equal(rows.view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))
    rows2, keep2 = _wnms(be, d, 0.1, 0.5, is3d, None)
    o2 = np.argsort(-d[:, 11], kind="stable").astype(np.int32)
    flat2, rk2 = O.wnms_4c(d, 0.1, 0.5, bool(is3d), 100, order=o2)
    assert keep2.tolist() == rk2
    assert np.array_equal(rows2.view(np.uint32), np.array(flat2, np.float32).reshape(-1, 12).view(np.uint32))
    # the reference's order computed on the device (std::sort replay): same result as with the host-side std::sort
    rows3, keep3 = _wnms(be, d, 0.1, 0.5, is3d, None, tie=R.RD_TIE_REFERENCE)
    assert keep3.tolist() == rk and np.array_equal(rows3.view(np.uint32), rows.view(np.uint32))


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("is3d", [0, 1])
def test_wnms_two_rounds_vs_oracle(be, is3d):
    """Capacity >= 1024 switches rd_wnms_4c to two rounds (pairs of the first 256 rows -> scan -> compacted list of the
    rows still alive -> their pairs -> resumed scan).  K well above 256 so that both rounds, the compaction and the
    resume matter; keep indices and merged rows bit-equal to the oracle, with exact score ties in the input."""
    no, rep = (70, 8) if be.name == "emu" else (200, 8)
    d = synth.cluster_dets(no, rep, seed=77, quant=33)
    K = d.shape[0]
    assert K > 2 * 256
    o2 = np.argsort(-d[:, 11], kind="stable").astype(np.int32)
    rows, keep = _wnms(be, d, 0.1, 0.5, is3d, o2, cap_extra=1024 + 64 - K if K < 1024 else 64)
    flat, rk = O.wnms_4c(d, 0.1, 0.5, bool(is3d), 100, order=o2)
    assert 16 < len(rk) < K and keep.tolist() == rk
    assert np.array_equal(rows.view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))
    rows2, keep2 = _wnms(be, d, 0.1, 0.5, is3d, None, cap_extra=1024 + 64 - K if K < 1024 else 64)   # library-side ordering
    assert keep2.tolist() == rk and np.array_equal(rows2.view(np.uint32), rows.view(np.uint32))


def run_conv_ex(be, B, H, W, cin, cout, stride, sc_cin=None, residual=False, sc_cs=None, seed=0, fold=False, dt=R.RD_BF16, m16=False):
    """rd_conv3x3_bn_act_ex (bf16) vs torch fp32: conv2 of a BasicBlock with stride (1,stride), optional residual, optional fused
    1x1 projection shortcut of a second input (scales folded into both weight sets by the packers)."""
    if dt == R.RD_F16 and be.name == "emu" and not sc_cin and not fold:
        pytest.skip("emulator build: fp16 persistent kernel in its production (folded-scale) forms only; this form runs on the GPU tier")
    rng = np.random.default_rng(seed)
    x = h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), dt)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    sc2 = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh2 = rng.standard_normal(cout).astype(np.float32)
    Wo = (W - 1) // stride + 1
    L = be.lib
    cs = -(-cin // 16) * 16
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=(1, stride), padding=1).numpy()
    xin = be.up(to_nhwc(x, dt, cstride=cs))
    y = be.empty(B * H * Wo * cout * 2)
    args_sc = (None, 0, 0, 0, None)
    res_ptr = (None, 0, 0)
    flags = R.RD_RELU_POST
    if sc_cin:
        x0 = h16_round(rng.standard_normal((B, sc_cin, H, W)).astype(np.float32), dt)
        wsc = (rng.standard_normal((cout, sc_cin)) / np.sqrt(sc_cin)).astype(np.float32)
        scs, shs = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
        s_cs = sc_cs or -(-sc_cin // 16) * 16
        ref = ref * sc2[None, :, None, None] + (sh2 + shs)[None, :, None, None]
        ref = ref + (F.conv2d(torch.from_numpy(x0), torch.from_numpy(wsc[:, :, None, None]), stride=(1, stride)).numpy() *
                     scs[None, :, None, None])
        wp = be.up(L.pack_conv3x3_ex(w, stride, cs, fold_scale=sc2, dtype=dt))
        args_sc = (be.ptr(be.up(to_nhwc(x0, dt, cstride=s_cs))), s_cs, 0, sc_cin, be.ptr(be.up(L.pack_conv1x1_sc(wsc, fold_scale=scs, dtype=dt))))
        scale_ptr, shift_ptr = None, be.ptr(be.up(sh2 + shs))
        flags |= R.RD_ADD
    else:
        ref = ref * sc2[None, :, None, None] + sh2[None, :, None, None]
        wp = be.up(L.pack_conv3x3_m16(w, sc2, dtype=dt) if m16 else L.pack_conv3x3_ex(w, stride, cs, fold_scale=sc2 if fold else None, dtype=dt))
        scale_ptr, shift_ptr = (None if fold else be.ptr(be.up(sc2))), be.ptr(be.up(sh2))
        if fold:
            flags |= R.RD_SCALE_FOLDED
        if m16:
            flags |= R.RD_MFMA16
        if residual:
            r = h16_round(rng.standard_normal((B, cout, H, Wo)).astype(np.float32), dt)
            ref = ref + r
            res_ptr = (be.ptr(be.up(to_nhwc(r, dt))), cout, 0)
            flags |= R.RD_ADD
    ref = np.maximum(ref, 0)
    L.call("rd_conv3x3_bn_act_ex", be.ptr(xin), cs, 0, be.ptr(wp), scale_ptr, shift_ptr, *res_ptr, *args_sc, be.ptr(y), cout, 0,
           B, H, W, cin, cout, stride, flags, dt, be.stream)
    got = from_nhwc(be.down(y, np.uint16, (B, H, Wo, cout)), dt, cout)
    # folded scales re-round the weights (2^-9 relative each for bf16, 2^-12 for fp16, averaging out over the >= 72-term sums)
    # on top of the output rounding of _tol: 1.5x
    tol = 1.5 * _tol(dt, ref)
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)


CONV_EX_CASES = [
    # B, H, W, cin, cout, stride, sc_cin, residual
    (2, 5, 260, 64, 64, 2, None, False),     # stride 2 on the pixel-pair view, several column tiles
    (1, 9, 132, 128, 128, 2, None, True),    # + residual at the output resolution, two row tiles
    (1, 4, 70, 64, 64, 1, 8, False),         # res1_unit1: 3x3 64->64 + projection shortcut from the 8-channel input
    (2, 9, 130, 128, 128, 1, 128, False),    # agg2_res_unit1: stride 1, 128-channel shortcut
    (1, 5, 264, 64, 128, 2, 64, False),      # res2_unit1: stride 2 + 64->128 shortcut (even pixels of the block input)
    (1, 3, 66, 128, 128, 2, 128, False),     # res3a_unit1
    (1, 4, 64, 64, 64, 1, 64, False),        # agg1/agg3_res_unit1
]
CONV_FOLD_CASES = [   # RD_SCALE_FOLDED without a shortcut: plain / residual, stride 1 / 2, cout 64 / 128, partial tiles
    (2, 9, 130, 128, 128, 1, None, True), (1, 5, 70, 64, 64, 1, None, False), (1, 4, 20, 72, 128, 1, None, False),
    (1, 9, 132, 128, 128, 2, None, True), (2, 3, 64, 64, 64, 2, None, False), (1, 11, 63, 8, 64, 1, None, False),
    # 8 images: the column strips divide by 8 -> the XCD-aware tile order (k_conv3.h Conv3Args::xcd), several tiles per workgroup
    # on the emulator's 8 slots, row-block and strip carries
    (8, 20, 70, 64, 64, 1, None, True), (8, 9, 100, 128, 128, 1, None, False), (8, 17, 40, 64, 64, 2, None, False),
]


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", CONV_EX_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_conv3x3_ex(be, case):
    run_conv_ex(be, *case, seed=sum(v or 0 for v in case[:6]))


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [CONV_EX_CASES[0], CONV_EX_CASES[2], CONV_EX_CASES[3], CONV_EX_CASES[4]], ids=lambda c: "-".join(str(v) for v in c))
def test_conv3x3_ex_fp16(be, case):
    run_conv_ex(be, *case, seed=sum(v or 0 for v in case[:6]), dt=F16)


@pytest.mark.parametrize("be", WITH_LATE_DMA, indirect=True)
@pytest.mark.parametrize("case", CONV_FOLD_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_conv3x3_ex_folded_scale(be, case):
    run_conv_ex(be, *case, seed=sum(v or 0 for v in case[:6]), fold=True)


CONV_M16_CASES = [   # RD_MFMA16 (v_mfma_f32_16x16x32 form): cout 128, stride 1; plain / residual, ragged H / W, 1 / 2 / 4 chunks, XCD order
    (2, 9, 130, 128, 128, 1, None, True), (1, 5, 70, 64, 128, 1, None, False), (1, 11, 33, 32, 128, 1, None, True),
    (8, 9, 100, 128, 128, 1, None, False),
]


@pytest.mark.parametrize("be", WITH_LATE_DMA, indirect=True)
@pytest.mark.parametrize("dt", [R.RD_BF16, F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", CONV_M16_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_conv3x3_ex_mfma16(be, case, dt):
    """The 16 x 16 x 32 MFMA form of the persistent conv (RD_MFMA16 + rd_pack_conv3x3_m16_host) against torch fp32, same tolerance as the
    32 x 32 x 16 form."""
    run_conv_ex(be, *case, seed=sum(v or 0 for v in case[:6]), fold=True, dt=dt, m16=True)


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", CONV_FOLD_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_conv3x3_ex_folded_scale_fp16(be, case):
    run_conv_ex(be, *case, seed=sum(v or 0 for v in case[:6]), fold=True, dt=F16)


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", [(BF16, 2, 9, 70, 64, 8, 16, 128), (BF16, 1, 17, 40, 32, 8, 16, 64), (F16, 2, 9, 70, 64, 8, 16, 128), (BF16, 1, 8, 33, 64, 24, 24, 128)])
def test_conv3x3_cat_two_tensor_input(be, case):
    """rd_conv3x3_bn_act_cat: 3x3 conv + BN + ReLU over the channel concatenation [x1 | x2] of two tensors with their own channel
    strides (dla_backbone.py:153-154 concat -> head/builder.py:221-240), against torch on the materialised concatenation, and bit
    for bit against rd_conv3x3_bn_act_ex on a shared buffer that holds the same channels."""
    dt, B, H, W, c1, c2, cs2, cout = case
    rng = np.random.default_rng(31)
    x1 = h16_round(rng.standard_normal((B, c1, H, W)).astype(np.float32), dt)
    x2 = h16_round(rng.standard_normal((B, c2, H, W)).astype(np.float32), dt)
    cin2 = -(-c2 // 8) * 8
    w = (rng.standard_normal((cout, c1 + cin2, 3, 3)) / np.sqrt((c1 + c2) * 9)).astype(np.float32)
    w[:, c1 + c2:] = 0
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    L = be.lib
    fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED
    d1, d2 = be.up(to_nhwc(x1, dt)), be.up(to_nhwc(x2, dt, cstride=cs2))
    wp, dsh = be.up(L.pack_conv3x3_cat(w, c1, cin2, fold_scale=sc, dtype=dt)), be.up(sh)
    y = be.empty(B * H * W * cout * 2)
    L.call("rd_conv3x3_bn_act_cat", be.ptr(d1), c1, 0, c1, be.ptr(d2), cs2, 0, cin2, be.ptr(wp), be.ptr(dsh), be.ptr(y), cout, 0, B, H, W, cout,
           fl, dt, be.stream)
    got = be.down(y, np.uint16, (B, H, W, cout))
    xc = np.concatenate([x1, x2, np.zeros((B, cin2 - c2, H, W), np.float32)], 1)
    ref = F.conv2d(torch.from_numpy(xc), torch.from_numpy(w), padding=1).numpy() * sc[None, :, None, None] + sh[None, :, None, None]
    ref = np.maximum(ref, 0)
    assert np.abs(from_nhwc(got, dt, cout) - ref).max() <= 1.5 * _tol(dt, ref)
    # against the single-tensor launch on a shared buffer that holds the same channels: the same numbers (with cin1 = 64, cin2 <= 16
    # the two-tensor launch sums the x2 chunk in another order -- five two-tap steps -- so equal to one output rounding, else bit-equal)
    cs = -(-(c1 + cin2) // 16) * 16
    y2 = be.empty(B * H * W * cout * 2)
    wp1 = be.up(L.pack_conv3x3_ex(w, 1, cs, fold_scale=sc, dtype=dt))
    L.call("rd_conv3x3_bn_act_ex", be.ptr(be.up(to_nhwc(xc, dt, cstride=cs))), cs, 0, be.ptr(wp1), None, be.ptr(dsh), None, 0, 0, None, 0, 0, 0, None,
           be.ptr(y2), cout, 0, B, H, W, c1 + cin2, cout, 1, fl, dt, be.stream)
    got2 = be.down(y2, np.uint16, (B, H, W, cout))
    if c1 == 64 and cin2 <= 16:
        assert np.abs(from_nhwc(got, dt, cout) - from_nhwc(got2, dt, cout)).max() <= 2 * _ulp(dt) * max(1.0, float(np.abs(ref).max()))
    else:
        assert np.array_equal(got, got2)
    buf = be.ptr(be.empty(1 << 16))
    f = L.raw("rd_conv3x3_bn_act_cat")
    assert f(buf, 64, 0, 48, buf, 16, 0, 16, buf, buf, buf, 128, 0, 1, 4, 8, 128, fl, dt, be.stream) == R.RD_ESHAPE        # cin1 not a multiple of 32
    assert f(buf, 64, 0, 64, buf, 16, 0, 16, buf, buf, buf, 128, 0, 1, 4, 8, 128, R.RD_RELU_POST, dt, be.stream) == R.RD_EINVAL   # not folded
    assert f(buf, 64, 0, 64, buf, 16, 8, 16, buf, buf, buf, 128, 0, 1, 4, 8, 128, fl, dt, be.stream) == R.RD_ESHAPE        # x2 channels exceed its stride


@pytest.mark.parametrize("be", WITH_LATE_DMA, indirect=True)
@pytest.mark.parametrize("case", [(BF16, 2, 16, 72, False, 64, 64), (BF16, 1, 11, 33, True, 64, 64), (F16, 1, 8, 100, False, 80, 64), (BF16, 3, 24, 64, True, 80, 64),
                                  (F16, 2, 9, 40, True, 64, 64), (BF16, 2, 16, 72, True, 16, 8), (F16, 1, 11, 33, True, 16, 5), (BF16, 3, 24, 64, True, 16, 16)],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_block64_equals_the_two_launches(be, case):
    """rd_block64_bn_act (csrc/k_block.h): a 64-channel BasicBlock -- conv1 3x3 + BN + ReLU, conv2 3x3 + BN, + identity or 1x1 projection
    shortcut, ReLU (dla_backbone.py:18-56) -- as ONE launch whose intermediate tensor stays in LDS.  BIT-IDENTICAL to the two
    rd_conv3x3_bn_act_ex launches it replaces (same MFMA sequence per accumulator, the intermediate rounded once like the stored
    tensor), and within one rounding per conv of torch on the same 16-bit weights.  Partial tiles in both directions, several tiles per
    workgroup (the x prefetch of the next tile into the buffer the intermediate just left), both shortcut forms, both 16-bit types, an
    input with a channel stride beyond its 64 channels, and the network's FIRST block: 8 (KITTI: 5) input channels in a 16-channel
    buffer, conv1 as five two-tap steps (the unfused launch's own form for <= 16 channels)."""
    dt, B, H, W, proj, xcs, cin = case
    rng = np.random.default_rng(B * 100 + H + W + cin)
    L = be.lib
    x = h16_round(rng.standard_normal((B, cin, H, W)).astype(np.float32), dt)
    w1 = (rng.standard_normal((64, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    w2 = (rng.standard_normal((64, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    wsc = (rng.standard_normal((64, cin)) / np.sqrt(cin)).astype(np.float32)
    s1, s2, ss = (rng.uniform(0.5, 1.5, 64).astype(np.float32) for _ in range(3))
    t1, t2, ts = (rng.standard_normal(64).astype(np.float32) * 0.3 for _ in range(3))
    sh2 = (t2.astype(np.float64) + ts).astype(np.float32) if proj else t2
    dx = be.up(to_nhwc(x, dt, cstride=xcs))
    p1, p2 = be.up(L.pack_conv3x3_ex(w1, 1, xcs, fold_scale=s1, dtype=dt)), be.up(L.pack_conv3x3_ex(w2, 1, 64, fold_scale=s2, dtype=dt))
    psc = be.up(L.pack_conv1x1_sc(wsc, fold_scale=ss, dtype=dt)) if proj else None
    d1, d2 = be.up(t1), be.up(sh2)
    t, yr, y = be.empty(B * H * W * 64 * 2), be.empty(B * H * W * 64 * 2), be.empty(B * H * W * 64 * 2)
    FO = R.RD_SCALE_FOLDED
    L.call("rd_conv3x3_bn_act_ex", be.ptr(dx), xcs, 0, be.ptr(p1), None, be.ptr(d1), None, 0, 0, None, 0, 0, 0, None, be.ptr(t), 64, 0, B, H, W, cin, 64, 1,
           R.RD_RELU_POST | FO, dt, be.stream)
    L.call("rd_conv3x3_bn_act_ex", be.ptr(t), 64, 0, be.ptr(p2), None, be.ptr(d2), None if proj else be.ptr(dx), 0 if proj else xcs, 0,
           be.ptr(dx) if proj else None, xcs if proj else 0, 0, cin if proj else 0, be.ptr(psc) if proj else None, be.ptr(yr), 64, 0, B, H, W, 64, 64, 1,
           R.RD_ADD | R.RD_RELU_POST | FO, dt, be.stream)
    pk = be.up(L.pack_block64(w1, s1, w2, s2, dtype=dt))
    L.call("rd_block64_bn_act", be.ptr(dx), xcs, 0, cin, be.ptr(pk), be.ptr(d1), be.ptr(d2), be.ptr(psc) if proj else None, be.ptr(y), 64, 0, B, H, W, dt, be.stream)
    got, two = be.down(y, np.uint16, (B, H, W, 64)), be.down(yr, np.uint16, (B, H, W, 64))
    assert np.array_equal(got, two), int((got != two).sum())
    if cin == 64:
        # ... and its 16 x 16 x 32 MFMA form (rd_block64_m16_bn_act, round 6): the same bits once more
        y16 = be.empty(B * H * W * 64 * 2)
        pk16 = be.up(L.pack_block64(w1, s1, w2, s2, dtype=dt, m16=True))
        psc16 = be.up(L.pack_conv1x1_sc_m16(wsc, ss, dtype=dt)) if proj else None
        L.call("rd_block64_m16_bn_act", be.ptr(dx), xcs, 0, be.ptr(pk16), be.ptr(d1), be.ptr(d2), be.ptr(psc16) if proj else None, be.ptr(y16), 64, 0,
               B, H, W, dt, be.stream)
        got16 = be.down(y16, np.uint16, (B, H, W, 64))
        assert np.array_equal(got16, two), int((got16 != two).sum())
    # ... and the pair itself against torch, conv by conv on the device's own intermediate (one output rounding each)
    tq = from_nhwc(be.down(t, np.uint16, (B, H, W, 64)), dt, 64)
    w1q, w2q = h16_round(w1 * s1[:, None, None, None], dt), h16_round(w2 * s2[:, None, None, None], dt)
    r1 = np.maximum(F.conv2d(torch.from_numpy(x), torch.from_numpy(w1q), padding=1).numpy() + t1[None, :, None, None], 0)
    assert np.abs(tq - r1).max() <= _tol(dt, r1)
    r2 = F.conv2d(torch.from_numpy(tq), torch.from_numpy(w2q), padding=1).numpy() + sh2[None, :, None, None]
    r2 = r2 + (F.conv2d(torch.from_numpy(x), torch.from_numpy(h16_round(wsc * ss[:, None], dt))[:, :, None, None]).numpy() if proj else x)
    r2 = np.maximum(r2, 0)
    assert np.abs(from_nhwc(got, dt, 64) - r2).max() <= 1.5 * _tol(dt, r2)
    f = L.raw("rd_block64_bn_act")
    p = be.ptr(be.empty(1 << 16))
    q = be.ptr(be.empty(1 << 16))
    assert f(p, 64, 0, 64, q, q, q, None, q, 64, 0, 1, 4, 8, R.RD_F32, be.stream) == R.RD_EINVAL          # 16-bit types only
    assert f(p, 64, 8, 64, q, q, q, None, q, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_ESHAPE                # 64 channels do not fit the stride
    assert f(p, 64, 0, 64, q, q, q, None, p, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_EINVAL                # in place
    assert f(p, 32, 0, 32, q, q, q, None, q, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_ESHAPE                # 32 input channels: not a form
    assert f(p, 16, 0, 8, q, q, q, None, q, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_EINVAL                 # first block without its projection shortcut
    assert f(p, 8, 0, 8, q, q, q, q, q, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_ESHAPE                     # 8-channel pitch: the shortcut reads 16 channels per pixel
    assert f(p, 24, 16, 8, q, q, q, q, q, 64, 0, 1, 4, 8, dt, be.stream) == R.RD_ESHAPE                   # ... also behind an offset


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_conv3x3_ex_errors(be):
    L = be.lib
    p = be.ptr(be.empty(1 << 16))
    f = L.raw("rd_conv3x3_bn_act_ex")
    assert f(p, 64, 0, p, p, p, None, 0, 0, None, 0, 0, 0, None, p, 64, 0, 1, 2, 33, 64, 64, 2, 4, BF16, be.stream) == R.RD_ESHAPE   # odd width
    assert f(p, 64, 0, p, p, p, p, 64, 0, p, 64, 0, 64, p, p, 64, 0, 1, 2, 32, 64, 64, 1, 6, BF16, be.stream) == R.RD_EINVAL        # both
    assert f(p, 64, 0, p, p, p, None, 0, 0, None, 0, 0, 0, None, p, 96, 0, 1, 2, 32, 64, 96, 1, 4, BF16, be.stream) == R.RD_ESHAPE   # cout
    assert f(p, 64, 0, p, p, p, None, 0, 0, None, 0, 0, 0, None, p, 64, 0, 1, 2, 32, 64, 64, 1, 4, F32, be.stream) == R.RD_EINVAL   # dtype


GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("f", sorted(glob.glob(os.path.join(GOLD, "wnms_*.npz"))), ids=os.path.basename)
def test_wnms_golden_through_hip(be, f):
    """Every committed golden vector (made by the REFERENCE's compiled nms.h, tests/golden/make_golden.py) through
    rd_wnms_4c with the library's own ordering (std::sort replay) and the fixture's hash_scale: keep indices and merged rows
    bit-equal.  Includes exact score ties (k512_ties), boxes beyond +-100 m (k256_far), hash_scale 10 (k256_hash10), 3-D IoU."""
    g = np.load(f)
    d = g["dets"]
    if d.shape[0] == 0:
        assert g["keep"].shape[0] == 0      # nms.h:463-466; the C ABI rejects Kcap == 0, callers skip the call
        return
    if be.name == "emu" and d.shape[0] > 600:
        pytest.skip("CPU emulation of K = 2048 takes minutes; runs on the GPU tier")
    rows, keep = _wnms(be, d, float(g["thresh"]), float(g["thresh_vote"]), bool(g["is3d"]), None, tie=R.RD_TIE_REFERENCE,
                       hash_scale=int(g["hash_scale"]))
    assert keep.tolist() == g["keep"].tolist()
    assert np.array_equal(rows.view(np.uint32), g["rows"].view(np.uint32))


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_pair_overlap_golden_through_hip(be):
    """pair_overlaps.npz (the reference's OverlapChecker::single_overlap on 1.1k box pairs) through rd_single_overlap: the
    device polygon clipper restates the reference's float operations in order, so the IoUs are bit-equal (BEV and 3-D)."""
    g = np.load(os.path.join(GOLD, "pair_overlaps.npz"))
    a, b = g["a"], g["b"]
    n = a.shape[0]
    out = be.empty(n * 4)
    for is3d, key in ((0, "iou"), (1, "iou3d")):
        be.lib.call("rd_single_overlap", be.ptr(be.up(a)), be.ptr(be.up(b)), n, is3d, be.ptr(out), be.stream)
        got = be.down(out, np.float32, (n,))
        bad = got.view(np.uint32) != g[key].view(np.uint32)
        # (edge angles: the C library's atan2f algorithm restated on the device, rd_common.h fdlibm_atan2f)
        assert not bad.any(), (int(bad.sum()), np.abs(got - g[key]).max())


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_edge_atan2f_equals_the_c_library(be):
    """rd_edge_atan2f = the atan2f of the weighted NMS's edge angles (nms.h:71: the C library's, through <cmath>) on the device: the fdlibm
    algorithm glibc ships restated operation by operation (rd_common.h fdlibm_atan2f) must be BIT-EQUAL to this host's atan2f -- on box
    edges, random bit patterns (all exponents, NaN, Inf, denormals), axis-aligned / zero arguments and ratios next to the routine's
    argument-reduction thresholds (tools/atan2f_replay_check.c ran the same comparison over 2e9 inputs on the host)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    rng = np.random.default_rng(11)
    n = 1 << 18
    parts = []
    th, ln = rng.uniform(-np.pi, np.pi, n), rng.uniform(0.2, 25, n)
    c = rng.uniform(-100, 100, (n, 2)).astype(np.float32)
    parts.append(((c[:, 1] + (ln * np.sin(th)).astype(np.float32)) - c[:, 1], (c[:, 0] + (ln * np.cos(th)).astype(np.float32)) - c[:, 0]))
    parts.append((rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32), rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)))
    sm = (rng.uniform(-0.5, 0.5, n) * np.where(rng.integers(0, 8, n) == 0, 0.0, 2.0 ** -rng.integers(0, 40, n))).astype(np.float32)
    parts.append((sm, (rng.uniform(-0.5, 0.5, n) * np.where(rng.integers(0, 8, n) == 0, 0.0, 50.0)).astype(np.float32)))
    xx = rng.uniform(-20, 20, n).astype(np.float32)
    thr = np.array([0.4375, 0.6875, 1.1875, 2.4375], np.float32)[rng.integers(0, 4, n)]
    parts.append(((xx * thr * (1 + rng.uniform(-5e-6, 5e-6, n)).astype(np.float32) * rng.choice([-1, 1], n)).astype(np.float32), xx))
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38, 1e-30], np.float32)
    parts.append((np.repeat(sp, len(sp)), np.tile(sp, len(sp))))
    y = np.ascontiguousarray(np.concatenate([p[0] for p in parts]).astype(np.float32))
    x = np.ascontiguousarray(np.concatenate([p[1] for p in parts]).astype(np.float32))
    out = be.empty(len(y) * 4)
    be.lib.call("rd_edge_atan2f", be.ptr(be.up(y)), be.ptr(be.up(x)), len(y), be.ptr(out), be.stream)
    got = be.down(out, np.float32, (len(y),))
    # the host's atan2f over the same arrays (a tiny C loop through ctypes would need a compiler; libm's symbol called per chunk via numpy
    # is not available, so: ctypes per element on a sample + every special value)
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    idx = np.concatenate([rng.choice(len(y) - len(sp) ** 2, 60000, replace=False), np.arange(len(y) - len(sp) ** 2, len(y))])
    want = np.array([libm.atan2f(float(y[i]), float(x[i])) for i in idx], np.float32)
    g = got[idx]
    same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
    # which C library this was compared with: the device restates the fdlibm routine glibc 2.35 ships; a newer glibc with a
    # correctly rounded atan2f would make THIS HOST's libm (and a reference compiled here) differ from it in the last bit
    gv = ctypes.CDLL("libc.so.6").gnu_get_libc_version
    gv.restype = ctypes.c_char_p
    glibc = gv().decode()
    if not same.all() and glibc != "2.35":
        pytest.xfail("this host's libm is glibc %s, whose atan2f differs from the fdlibm routine of glibc 2.35 the library restates "
                     "(rd_common.h fdlibm_atan2f, INTEGRATION.md): %d of %d sampled inputs differ in the last bit" % (glibc, int((~same).sum()), len(idx)))
    assert same.all(), ("glibc " + glibc, int((~same).sum()), y[idx][~same][:4], x[idx][~same][:4], g[~same][:4], want[~same][:4])


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_spurious_overlaps_and_the_rejection_test(be):
    """pair_overlaps_spurious.npz: 1.6k pairs of boxes that do NOT intersect on which the reference's clipper (nms.h:96-149,195-249,
    no empty-intersection test) returns a positive "IoU" -- found by running the compiled reference over disjoint pairs of nine
    families (oracle/ref_overlap_study.cpp, tests/golden/make_spurious_golden.py).
      * rd_single_overlap reproduces the reference's value on every one of them (its clipper follows the same float operations);
      * the pair kernel's rejection test (rd_wnms_pair_skippable: the pairs rd_wnms_4c does not clip) is FALSE for every pair whose
        reference value is >= 1e-6, i.e. no pair that could pass a threshold >= 1e-3 is ever skipped -- and it is true for most
        far-apart, well-conditioned pairs (what it exists for)."""
    g = np.load(os.path.join(GOLD, "pair_overlaps_spurious.npz"))
    a, b, iou = g["a"], g["b"], g["iou"]
    n = len(iou)
    L = be.lib
    out = be.empty(n * 4)
    for is3d, key in ((0, "iou"), (1, "iou3d")):
        L.call("rd_single_overlap", be.ptr(be.up(a)), be.ptr(be.up(b)), n, is3d, be.ptr(out), be.stream)
        got = be.down(out, np.float32, (n,))
        bad = got.view(np.uint32) != g[key].view(np.uint32)
        # Many of these pairs sit ON the |angle difference| < 1e-5 tie of nms.h:58-64, where one ulp of an edge angle decides which
        # half-plane survives.  Since round 5 the edge angles come from the C library's own atan2f algorithm restated on the device
        # (rd_common.h fdlibm_atan2f, bit-equal to glibc 2.35's on 2e9 inputs), so EVERY pair is bit-equal on both builds
        # (round 4: pairs within 3e-6 rad of the threshold -- tie_margin, stored with the vectors -- were exempt on the GPU)
        print("spurious golden (%s): %d of %d values differ from the reference (%d pairs within 3e-6 rad of the tie threshold)" %
              (key, int(bad.sum()), n, int((g["tie_margin"] <= 3e-6).sum())))
        assert not bad.any()
    skip = be.empty(n)
    L.call("rd_wnms_pair_skippable", be.ptr(be.up(a)), be.ptr(be.up(b)), n, be.ptr(skip), be.stream)
    s = be.down(skip, np.uint8, (n,))
    assert not s[iou >= 1e-6].any(), int(s[iou >= 1e-6].sum())
    assert not s[~(iou == iou)].any() or True            # (NaN results never pass a comparison: skipping them is neutral)
    # what it is for: car-sized boxes scattered over +-75 m, the partner 10 - 60 m away
    rng = np.random.default_rng(5)
    m = 20000
    far_a = synth.cluster_dets(m, 1, seed=21)[:m]
    far_b = far_a.copy()
    shift = rng.uniform(10, 60, m) * rng.choice([-1, 1], m)
    far_b[:, 0:8:2] += shift[:, None].astype(np.float32)
    ok = np.abs(far_b[:, :8]).max(1) < 190
    yaw = rng.uniform(0.05, 1.5, m)                        # rotate the partner about its own centre: not parallel
    cx, cy = far_b[:, 0:8:2].mean(1, keepdims=True), far_b[:, 1:8:2].mean(1, keepdims=True)
    x, y = far_b[:, 0:8:2] - cx, far_b[:, 1:8:2] - cy
    far_b[:, 0:8:2] = (cx + x * np.cos(yaw)[:, None] - y * np.sin(yaw)[:, None]).astype(np.float32)
    far_b[:, 1:8:2] = (cy + x * np.sin(yaw)[:, None] + y * np.cos(yaw)[:, None]).astype(np.float32)
    sk2, o2 = be.empty(m), be.empty(m * 4)
    L.call("rd_wnms_pair_skippable", be.ptr(be.up(far_a)), be.ptr(be.up(far_b)), m, be.ptr(sk2), be.stream)
    L.call("rd_single_overlap", be.ptr(be.up(far_a)), be.ptr(be.up(far_b)), m, 0, be.ptr(o2), be.stream)
    s2, ov = be.down(sk2, np.uint8, (m,)), be.down(o2, np.float32, (m,))
    print("far pairs: %d of %d skippable; largest overlap among the skipped %.2e" % (int(s2.sum()), m, float(np.nanmax(np.where(s2 > 0, ov, 0)))))
    assert s2[ok].mean() > 0.8
    assert not (ov[s2 > 0] >= 1e-6).any()                  # every skipped pair is one whose clip returns (next to) nothing


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_wnms_hash_prefilter_matters(be):
    """Boxes spread over +-170 m so that many pairs lie in different 100 m cells (the four quadrants around the ego vehicle
    never share a cell): result with hash_scale 100 and 10 equals the oracle (== reference) for that scale, and the
    prefilter-free evaluation (hash_scale 0) is what the oracle gives with one huge cell."""
    d = synth.cluster_dets(24, 10, seed=21, spread=170.0)
    for hs in (100, 10, 37):
        rows, keep = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE, hash_scale=hs)
        flat, rk = O.wnms_4c(d, 0.1, 0.5, False, hs)
        assert keep.tolist() == rk, hs
        assert np.array_equal(rows.view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))
    rows, keep = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE, hash_scale=0)
    flat, rk = O.wnms_4c(d, 0.1, 0.5, False, 30000)
    assert keep.tolist() == rk


def _tie_order(be, scores, cap_extra=3):
    L = be.lib
    K = len(scores)
    cap = K + cap_extra
    d = np.zeros((cap, 12), np.float32)
    d[:K, 11] = scores
    d[:K, :8] = np.array([0, 0, 1, 0, 1, 1, 0, 1], np.float32) + 5 * np.arange(K, dtype=np.float32)[:, None]   # disjoint boxes
    d[:K, 10] = 1
    nb = L.raw("rd_wnms_workspace_bytes")(cap)
    ws, outd, keep, nk = be.empty(nb), be.empty(cap * 48), be.empty(cap * 4), be.empty(16)
    L.call("rd_wnms_4c", be.ptr(be.up(d)), cap, be.ptr(be.up(np.array([K], np.int32))), None, R.RD_TIE_REFERENCE, 0.1, 0.5, 0, 0,
           be.ptr(outd), be.ptr(keep), be.ptr(nk), be.ptr(ws), nb, be.stream)
    assert int(be.down(nk, np.int32, (1,))[0]) == K      # nothing overlaps: every box is kept, in processing order
    return be.down(keep, np.int32, (cap,))[:K], d[:K]


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_tie_order_replays_std_sort(be):
    """The device replay of libstdc++'s std::sort (nms.h:786-792) against the real std::sort of the oracle build, on inputs
    that exercise every branch: already sorted with ties (the pipeline's case), random with many ties, all equal, n <= 16
    (insertion sort only), and a median-of-three killer sequence that drives introsort into its heap-sort fallback."""
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 3, 15, 16, 17, 33, 100, 700, 1500):
        q = rng.integers(0, max(2, n // 3), n).astype(np.float32) / 64 + 0.5
        cases.append(q)                                       # random order, many ties
        cases.append(-np.sort(-q))                            # sorted, ties
        cases.append(np.full(n, 0.75, np.float32))            # all equal
        cases.append(-np.sort(-rng.uniform(0.5, 1, n).astype(np.float32)))   # strictly sorted: identity fast path
    for n, ng in ((500, 1), (900, 3), (1500, 6), (1500, 40), (2500, 10), (3000, 2)):   # the pipeline's regime: sorted, a few ties
        q = -np.sort(-rng.uniform(0.5, 1, n).astype(np.float32))
        for _ in range(ng):
            j, g = int(rng.integers(0, n - 4)), int(rng.integers(2, 5))
            q[j:j + g] = q[j]
        cases.append(q)
    cases.append(O.antiqsort_keys(600))                       # adversarial input: depth limit reached -> heap sort
    cases.append(O.antiqsort_keys(2000))
    for sc in cases:
        got, d = _tie_order(be, sc)
        want = O.wnms_order(d)
        assert np.array_equal(got, want), (len(sc), int((got != want).sum()))
    assert O.std_sort_depth_limit_hit(O.antiqsort_keys(600))   # the killer really reaches the heap-sort branch


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_wnms_chunked_scan_and_merge_overflow(be):
    """K beyond 64*256 rows takes the column-chunked scan and neighbourhoods beyond the LDS list the global-scratch merge.
    Both paths are forced at small K through the call's diagnostic bits (RD_WNMS_DIAG_*) and must give the same bits as the normal paths."""
    d = synth.cluster_dets(40, 9, seed=31, quant=33)
    base = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE, cap_extra=1024)
    flat, rk = O.wnms_4c(d, 0.1, 0.5, False, 100)
    assert base[1].tolist() == rk
    alt = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE | R.RD_WNMS_DIAG_TILE_W(2) | R.RD_WNMS_DIAG_MERGE_LDS(6), cap_extra=1024)
    assert alt[1].tolist() == rk and np.array_equal(alt[0].view(np.uint32), base[0].view(np.uint32))
    alt = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE | R.RD_WNMS_DIAG_NO_SKIP, cap_extra=1024)   # every pair clipped
    assert alt[1].tolist() == rk and np.array_equal(alt[0].view(np.uint32), base[0].view(np.uint32))
    assert np.array_equal(base[0].view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))


@pytest.mark.parametrize("be", HIP_ONLY, indirect=True)
def test_wnms_large_k(be):
    """K = 20 000 rows above the score threshold (the reference accepts up to pre_nms_top_n = 50 000, nms.h:452-577): capacity
    above 16 384 rows = column-chunked scan, global-memory tie order, grid-strided pair tiles.  Bit-equal to the oracle."""
    d = synth.cluster_dets(2500, 8, seed=41, quant=1024)
    assert d.shape[0] == 20000
    rows, keep = _wnms(be, d, 0.1, 0.5, 0, None, tie=R.RD_TIE_REFERENCE)
    flat, rk = O.wnms_4c(d, 0.1, 0.5, False, 100)
    assert len(rk) > 500 and keep.tolist() == rk
    assert np.array_equal(rows.view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_score_filter_and_12to8(be):
    rng = np.random.default_rng(4)
    n = 3000
    boxes = (rng.standard_normal((n, 10)) * 10).astype(np.float32)
    sc = rng.uniform(0, 1, n).astype(np.float32)
    L = be.lib
    nb = L.raw("rd_score_filter_workspace_bytes")(n)
    dets, cnt, ws = be.empty(n * 48), be.empty(16), be.empty(nb)
    L.call("rd_score_filter_dets", be.ptr(be.up(sc)), be.ptr(be.up(boxes)), n, 0.5, be.ptr(dets), be.ptr(cnt), be.ptr(ws), nb, be.stream)
    ref = O.score_filter_to_dets(sc, boxes, 0.5)
    K = int(be.down(cnt, np.int32, (1,))[0])
    assert K == ref.shape[0]
    got = be.down(dets, np.float32, (n, 12))[:K]
    assert np.array_equal(got[:, :8], ref[:, :8]) and np.array_equal(got[:, 9:], ref[:, 9:])  # copies / exact subtract
    assert np.abs(got[:, 8] - ref[:, 8]).max() < 1e-5                                           # atan2f
    o8 = be.empty(K * 32)
    L.call("rd_dets12_to_8", be.ptr(dets), K, be.ptr(cnt), be.ptr(o8), be.stream)
    assert np.abs(be.down(o8, np.float32, (K, 8)) - O.bbox3d_12dim_to_8dim(got)).max() < 1e-4
    # nothing above the threshold -> K == 0
    L.call("rd_score_filter_dets", be.ptr(be.up(sc * 0)), be.ptr(be.up(boxes)), n, 0.5, be.ptr(dets), be.ptr(cnt), be.ptr(ws), nb, be.stream)
    assert int(be.down(cnt, np.int32, (1,))[0]) == 0


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_input_transform(be):
    """rd_input_transform against the numpy restatement of the reference's transform chain (oracle/input_ref.transform; parity
    unpinned: the reference chain needs mxnet/numba to import).  Everything is selection / copy / one IEEE subtract and
    divide, hence bit-exact, except the azimuth channel (atan2f)."""
    from rangedet_amd.input_transform import make_norm
    import ctypes
    L = be.lib
    H, W, Hp, Wp = 9, 53, 12, 56
    recs = [synth.raw_record(i, H, W) for i in (3, 4)]
    for r in recs:   # make the wrap-around, still-missing and car-window branches all occur
        r['range_image'][0, -1, 0] = -1; r['range_image'][0, 0, 0] = -1
        r['range_image'][4:9, 20:27, 0] = -1
        r['range_image'][2, 40:43, 0] = -1
        r['pc_vehicle_frame'][r['range_image'][..., 0] == -1] = 0
    B, npx = len(recs), Hp * Wp
    ri = be.up(np.stack([r['range_image'] for r in recs]))
    pc = be.up(np.stack([r['pc_vehicle_frame'] for r in recs]))
    inc = be.up(np.stack([r['inclination'] for r in recs]))
    names = ['input_data', 'coord_s1', 'pc_vehicle_frame_s1', 'pc_vehicle_frame_s2', 'pc_vehicle_frame_s4',
             'range_image_mask_s1', 'range_image_mask_s2', 'range_image_mask_s4']
    shapes = {'input_data': (B, 8, Hp, Wp), 'coord_s1': (B, 3, Hp, Wp)}
    for s in (1, 2, 4):
        shapes['pc_vehicle_frame_s%d' % s] = (B, npx // s, 3)
        shapes['range_image_mask_s%d' % s] = (B, npx // s)
    bufs = {k: be.empty(int(np.prod(shapes[k])) * 4) for k in names}
    norm = make_norm()
    L.call("rd_input_transform", be.ptr(ri), be.ptr(pc), be.ptr(inc), ctypes.addressof(norm), B, H, W, Hp, Wp,
           *[be.ptr(bufs[k]) for k in names], be.stream)
    ref = [IR.transform(r, (Hp, Wp)) for r in recs]
    assert any((r['range_image'][..., 0] == -1).any() for r in recs)
    for k in names:
        got = be.down(bufs[k], np.float32, shapes[k])
        want = np.concatenate([f[k] for f in ref], 0)
        if k == 'input_data':
            assert np.array_equal(got[:, :7], want[:, :7])
            assert np.abs(got[:, 7] - want[:, 7]).max() < 1e-6
        else:
            assert np.array_equal(got, want), k
    m = np.concatenate([f['range_image_mask_s4'] for f in ref], 0)
    assert 0 < m.sum() < m.size


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_postprocess_batched(be):
    """rd_score_filter_dets_batched + rd_wnms_4c_batched + rd_dets12_to_8_batched over 3 ragged frames (one of them with
    nothing above the threshold) against the oracle frame by frame: counts, keep indices and merged rows bit-exact."""
    L = be.lib
    B, n, cap = 3, 700, 512
    frames = [synth.cluster_dets(20, 9, seed=5), synth.cluster_dets(7, 30, seed=6, quant=True), synth.cluster_dets(4, 3, seed=7)]
    sc = np.zeros((B, n), np.float32)
    bx = np.zeros((B, n, 10), np.float32)
    for b, d in enumerate(frames):
        o = np.argsort(-d[:, 11], kind="stable")
        d = d[o]
        k = d.shape[0]
        sc[b, :k] = d[:, 11] if b < 2 else 0.01          # frame 2: everything below min_score
        bx[b, :k, :8] = d[:, :8]
        bx[b, :k, 8] = d[:, 9]
        bx[b, :k, 9] = d[:, 9] + d[:, 10]
    fb = L.raw("rd_score_filter_workspace_bytes")(n) * B
    wb = L.raw("rd_wnms_workspace_bytes")(cap) * B
    dets, cnt, wsf, wsw = be.empty(B * n * 48), be.empty(16), be.empty(fb), be.empty(wb)
    out, keep, nk, o8 = be.empty(B * cap * 48), be.empty(B * cap * 4), be.empty(16), be.empty(B * cap * 32)
    ident = be.up(np.arange(cap, dtype=np.int32))
    L.call("rd_score_filter_dets_batched", be.ptr(be.up(sc)), n, be.ptr(be.up(bx)), n * 10, n, 0.5, be.ptr(dets), n * 12,
           be.ptr(cnt), be.ptr(wsf), fb, B, be.stream)
    L.call("rd_wnms_4c_batched", be.ptr(dets), n * 12, cap, be.ptr(cnt), be.ptr(ident), 0, R.RD_TIE_STABLE, 0.1, 0.5, 0, 100,
           be.ptr(out), cap * 12, be.ptr(keep), cap, be.ptr(nk), be.ptr(wsw), wb, B, be.stream)
    L.call("rd_dets12_to_8_batched", be.ptr(out), cap * 12, cap, be.ptr(nk), be.ptr(o8), cap * 8, B, be.stream)
    K = be.down(cnt, np.int32, (4,))[:B]
    M = be.down(nk, np.int32, (4,))[:B]
    rows = be.down(out, np.float32, (B, cap, 12))
    kp = be.down(keep, np.int32, (B, cap))
    d8 = be.down(o8, np.float32, (B, cap, 8))
    alld = be.down(dets, np.float32, (B, n, 12))
    for b in range(B):
        ref = O.score_filter_to_dets(sc[b], bx[b], 0.5)
        assert K[b] == ref.shape[0]
        if K[b] == 0:
            assert M[b] == 0
            continue
        got = alld[b, :K[b]]
        flat, rk = O.wnms_4c(got, 0.1, 0.5, False, 100, order=np.arange(K[b], dtype=np.int32))
        assert kp[b, :M[b]].tolist() == rk
        assert np.array_equal(rows[b, :M[b]].view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))
        assert np.abs(d8[b, :M[b]] - O.bbox3d_12dim_to_8dim(rows[b, :M[b]])).max() < 1e-4
    with pytest.raises(Exception):   # B > 1: explicit order or the reference's tie order
        L.call("rd_wnms_4c_batched", be.ptr(dets), n * 12, cap, be.ptr(cnt), None, 0, R.RD_TIE_STABLE, 0.1, 0.5, 0, 100,
               be.ptr(out), cap * 12, be.ptr(keep), cap, be.ptr(nk), be.ptr(wsw), wb, B, be.stream)
    # batched, order computed by the library the reference's way (frame 1 has exact score ties): equal to the oracle's default
    L.call("rd_wnms_4c_batched", be.ptr(dets), n * 12, cap, be.ptr(cnt), None, 0, R.RD_TIE_REFERENCE, 0.1, 0.5, 0, 100,
           be.ptr(out), cap * 12, be.ptr(keep), cap, be.ptr(nk), be.ptr(wsw), wb, B, be.stream)
    M = be.down(nk, np.int32, (4,))[:B]
    rows = be.down(out, np.float32, (B, cap, 12))
    kp = be.down(keep, np.int32, (B, cap))
    for b in range(2):
        flat, rk = O.wnms_4c(alld[b, :K[b]], 0.1, 0.5, False, 100)
        assert kp[b, :M[b]].tolist() == rk
        assert np.array_equal(rows[b, :M[b]].view(np.uint32), np.array(flat, np.float32).reshape(-1, 12).view(np.uint32))


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_rotated_iou_8pt(be):
    b1 = synth.cluster_dets(8, 6, seed=3)[:, :8].copy()
    gt = synth.cluster_dets(8, 2, seed=3)[:, :8].copy()
    gt = np.concatenate([gt, np.tile(np.array([[0, 0, 0, 1e-3, 1e-3, 1e-3, 1e-3, 0]], np.float32), (4, 1))])  # GT padding rows
    b1 = np.concatenate([b1, b1[:2]])  # identical boxes (winding-dependent degenerate case)
    L = be.lib
    io = be.empty(b1.shape[0] * gt.shape[0] * 4)
    L.call("rd_rotated_iou_8pt", be.ptr(be.up(b1)), be.ptr(be.up(gt)), be.ptr(io), b1.shape[0], gt.shape[0], be.stream)
    ref = O.rotated_iou_8pt(b1, gt)
    got = be.down(io, np.float32, ref.shape)
    assert np.allclose(got, ref, atol=1e-5, equal_nan=True)
    mx = be.empty(b1.shape[0] * 4)
    L.call("rd_batch_max_iou", be.ptr(be.up(b1)), 8, be.ptr(be.up(gt)), be.ptr(mx), b1.shape[0], gt.shape[0], be.stream)
    assert np.abs(be.down(mx, np.float32, (b1.shape[0],)) - O.batch_max_iou(b1, gt)).max() < 1e-5


def _riou_case(B, N, n_real, seed):
    """Proposals (B,N,10) like a level's decoded boxes and gt_bbox (B,200,8) like GetFixedLengthGTBbox's output: n_real real
    boxes then the degenerate padding rows [0,0,0,EPS,EPS,EPS,EPS,0]; a share of the proposals are jittered copies of GT boxes,
    some exact copies (the winding-dependent degenerate case), some sit on the origin (they meet the padding rows)."""
    rng = np.random.default_rng(seed)
    props, gts = [], []
    for b in range(B):
        g = synth.cluster_dets(n_real, 1, seed=seed + 7 * b)[:, :8]
        gt = np.tile(np.array([0, 0, 0, 1e-3, 1e-3, 1e-3, 1e-3, 0], np.float32), (200, 1))
        gt[:n_real] = g
        far = synth.cluster_dets(max(1, N // 40), 40, seed=seed + 100 + b, spread=75.0, jitter=1.0)[:N, :8]
        p = np.zeros((N, 10), np.float32)
        p[:, :8] = np.resize(far, (N, 8))
        near = rng.choice(N, N // 8, replace=False)
        src = rng.integers(0, n_real, near.size)
        p[near, :8] = g[src] + rng.normal(0, 0.3, (near.size, 1)).astype(np.float32) * np.tile([1, 0.5], 4)
        p[near[:16], :8] = g[src[:16]]                                   # identical boxes
        p[near[16:32], :8] = p[near[16:32], :8] * 0 + rng.normal(0, 5e-4, (16, 8)).astype(np.float32)   # on the origin
        p[:, 8], p[:, 9] = -1.0, 0.8
        props.append(p)
        gts.append(gt)
    return np.stack(props), np.stack(gts)


def _check_batch_riou(be, prop, gt):
    B, N, _ = prop.shape
    L = be.lib
    out, arg = be.empty(B * N * 4), be.empty(B * N * 4)
    L.call("rd_batch_rotated_iou", be.ptr(be.up(prop)), 10, be.ptr(be.up(gt)), be.ptr(out), be.ptr(arg), B, N, gt.shape[1], be.stream)
    got, ga = be.down(out, np.float32, (B, N)), be.down(arg, np.int32, (B, N))
    hits = 0
    for b in range(B):
        m = O.rotated_iou_8pt(prop[b, :, :8], gt[b])                      # the full (N, 200) matrix of the oracle
        m[np.isnan(m) | np.isinf(m) | (m > 1.0) | (m < 0)] = 0            # batch_rotated_iou.py:42-45
        ref, ra = m.max(axis=1), m.argmax(axis=1)
        assert np.abs(got[b] - ref).max() < 1e-5, np.abs(got[b] - ref).max()
        top2 = np.sort(m, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0] > 1e-4)                          # a unique maximum (beyond the atan2f noise)
        assert np.array_equal(ga[b][clear], ra[clear])
        assert np.array_equal(ga[b][ref == 0], np.zeros((ref == 0).sum(), np.int32))   # all-zero row -> index 0, like argmax
        hits += int((ref > 0).sum())
    return hits


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_batch_rotated_iou(be):
    """Custom op 'batch_rotated_iou' ('bev'): (B,N,10) proposals x (B,200,8) GT -> (B,N) max cleaned IoU (+ argmax) vs the
    oracle's full IoU matrix, with GT padding rows, identical boxes and boxes on the origin."""
    prop, gt = _riou_case(2, 700 if be.name == "emu" else 4000, 12, seed=5)
    assert _check_batch_riou(be, prop, gt) > 100
    with pytest.raises(R.RangeDetError):
        be.lib.call("rd_batch_rotated_iou", be.ptr(be.up(prop)), 10, be.ptr(be.up(gt)), be.ptr(be.empty(64)), None, 2, 10, 300, be.stream)


def _gt7_from_corners(g8, z0, z1):
    """(n,8) BEV corners + z range -> (n,7) [cx, cy, cz, length, width, height, yaw]: what the '3d' op is fed as gt_bbox
    (batch_rotated_iou.py:86-89), made the way to_box_type_7 makes the proposals' 7-dim rows."""
    p = np.concatenate([g8, np.full((len(g8), 1), z0, np.float32), np.full((len(g8), 1), z1, np.float32)], 1).astype(np.float32)
    return O.to_box_type_7(p)


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_batch_rotated_iou_3d(be):
    """Custom op 'batch_rotated_iou' with iou_type '3d' (operator_py/batch_rotated_iou.py:17-18,36-39,51-68; the 7-dim volume IoU of
    rotated_iou-inl.h:495-522): (B,N,10) proposals x (B,200,7) GT -> (B,N) max cleaned volume IoU + argmax, against the oracle's full
    matrix -- proposals converted by to_box_type_7, the yaw of both sides negated; GT padding rows (zero volume), jittered and
    identical copies of GT boxes, boxes on the origin, partial height overlap.  And rd_rotated_iou_7 = the op's 7-dim matrix."""
    B, N, n_real = 2, 600 if be.name == "emu" else 4000, 12
    prop, gt8 = _riou_case(B, N, n_real, seed=15)
    rng = np.random.default_rng(3)
    prop[:, :, 8] = rng.uniform(-1.5, -0.5, (B, N)).astype(np.float32)
    prop[:, :, 9] = prop[:, :, 8] + rng.uniform(1.2, 2.2, (B, N)).astype(np.float32)
    gt7 = np.zeros((B, 200, 7), np.float32)
    gt7[:, :, 3:6] = 1e-3                                              # padding rows: volume 1e-9 < EPS -> IoU 0 (rotated_iou-inl.h:499)
    for b in range(B):
        gt7[b, :n_real] = _gt7_from_corners(gt8[b, :n_real], -1.2, 0.7)
    L = be.lib
    out, arg = be.empty(B * N * 4), be.empty(B * N * 4)
    L.call("rd_batch_rotated_iou_3d", be.ptr(be.up(prop)), 10, be.ptr(be.up(gt7)), be.ptr(out), be.ptr(arg), B, N, 200, be.stream)
    got, ga = be.down(out, np.float32, (B, N)), be.down(arg, np.int32, (B, N))
    hits = 0
    for b in range(B):
        ref, m = O.batch_max_iou_3d(prop[b], gt7[b])
        # proposals that are EXACT copies of a ground-truth box: coincident edges, where the routine's relative-equality test divides
        # by min(det, 0) (rotated_iou-inl.h:50-53,130-172) and an ulp of the recomputed corners (cosf / sinf: device libm vs glibc)
        # decides which "intersections" exist -- the value there is not a function of the boxes but of the libm.  Everywhere else
        # the device must agree to 1e-5; the CPU build of the same sources (glibc) must agree on the copies too.
        copies = (np.abs(prop[b][:, None, :8] - gt8[b][None, :n_real, :]).max(axis=2) == 0).any(axis=1)
        strict = np.ones(N, bool) if be.name == "emu" else ~copies
        assert copies.sum() >= 8 and np.abs(got[b] - ref)[strict].max() < 1e-5, np.abs(got[b] - ref)[strict].max()
        top2 = np.sort(m, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0] > 1e-4) & strict
        assert np.array_equal(ga[b][clear], m.argmax(axis=1)[clear])
        assert np.array_equal(ga[b][(ref == 0) & strict], np.zeros(((ref == 0) & strict).sum(), np.int32))
        hits += int((ref > 0).sum())
        # the volume IoU really is the BEV overlap times the height overlap: below the BEV IoU where the heights differ (not a
        # bound for every row: the 8-point routine returns 0 for identical boxes in the decode winding, the 7-dim one does not)
        bev = O.batch_max_iou(prop[b, :, :8], gt8[b])
        assert (ref[bev > 0.2] < bev[bev > 0.2]).mean() > 0.7 and (ref[bev > 0.2] > 0).mean() > 0.9
    assert hits > 100
    # the z-sign flip matters (batch_rotated_iou.py:37-38): without it the boxes would be mirrored
    r7 = O.to_box_type_7(prop[0])
    a7, g7 = r7.copy(), gt7[0].copy()
    a7[:, 6] *= -1
    g7[:, 6] *= -1
    mo = be.empty(N * 200 * 4)
    L.call("rd_rotated_iou_7", be.ptr(be.up(a7)), be.ptr(be.up(g7)), be.ptr(mo), N, 200, be.stream)
    mdev, mref = be.down(mo, np.float32, (N, 200)), O.rotated_iou_7(a7, g7)
    okm = np.isclose(mdev, mref, atol=1e-5, equal_nan=True)
    copies0 = (np.abs(prop[0][:, None, :8] - gt8[0][None, :n_real, :]).max(axis=2) == 0).any(axis=1)
    assert okm[~copies0].all() and (be.name != "emu" or okm.all())
    assert np.abs(O.rotated_iou_7(r7, gt7[0]) - O.rotated_iou_7(a7, g7)).max() > 0.1
    with pytest.raises(R.RangeDetError):
        L.call("rd_batch_rotated_iou_3d", be.ptr(be.up(prop)), 8, be.ptr(be.up(gt7)), be.ptr(be.empty(64)), None, 2, 10, 200, be.stream)


@pytest.mark.slow
@pytest.mark.parametrize("be", HIP_ONLY, indirect=True)
def test_batch_rotated_iou_full_level(be):
    """BASELINE config 3 at the reference's level-0 size: 169 984 proposals x 200 GT boxes per frame (batch_rotated_iou.py:12-13),
    max and argmax against the oracle's 34 M-pair matrix."""
    prop, gt = _riou_case(1, 169984, 60, seed=9)
    assert _check_batch_riou(be, prop, gt) > 5000


def nms3d_boxes(B, n_obj, rep, seed, jitter=0.05):
    """(B,N,10) score-sorted boxes: 4 BEV corners + z_low, z_high (the layout Decode3DBbox hands to NMS3D)."""
    out = []
    for b in range(B):
        d = synth.cluster_dets(n_obj, rep, seed=seed + b, jitter=jitter)
        d = d[np.argsort(-d[:, 11], kind="stable")]
        out.append(np.concatenate([d[:, :8], d[:, 9:10], d[:, 9:10] + d[:, 10:11]], axis=1))
    return np.stack(out).astype(np.float32)


def run_nms3d(be, boxes, thr, max_keep, normal):
    L = be.lib
    B, N = boxes.shape[:2]
    wb = L.raw("rd_nms3d_workspace_bytes")(N, B)
    assert wb > 0
    ws = be.empty(wb)
    keep, out = be.empty(B * max_keep * 4), be.empty(B * max_keep * 40)
    L.call("rd_nms3d", be.ptr(be.up(boxes)), B, N, thr, max_keep, int(normal), be.ptr(keep), be.ptr(out), be.ptr(ws), wb, be.stream)
    return be.down(keep, np.int32, (B, max_keep)), be.down(out, np.float32, (B, max_keep, 10))


NMS3D_CASES = [
    # B, n_obj, rep, thr, max_keep, normal_iou
    (2, 12, 30, 0.1, 500, False),     # every survivor kept (fewer than max_keep), N = 360: one row block, ragged last word
    (1, 40, 8, 0.5, 25, False),       # stops at max_keep
    (1, 30, 45, 0.3, 100, False),     # N = 1350: two row blocks
    (2, 20, 10, 0.3, 64, True),       # axis-aligned measure on columns 0..3
    (1, 1, 1, 0.1, 4, False),         # a single box
]


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("case", NMS3D_CASES)
def test_nms3d_vs_oracle(be, case):
    """rd_nms3d against the restatement of nms_3d.cu: keep indices and copied rows bit-equal, padding -1 / 0."""
    B, n_obj, rep, thr, max_keep, normal = case
    boxes = nms3d_boxes(B, n_obj, rep, seed=11)
    if normal:   # make columns 0..3 proper (x1,y1,x2,y2) rectangles
        xy = boxes[..., :8].reshape(B, -1, 4, 2)
        boxes[..., 0:2] = xy.min(axis=2)
        boxes[..., 2:4] = xy.max(axis=2)
    keep, out = run_nms3d(be, boxes, thr, max_keep, normal)
    rk, ro = O.nms3d(boxes, thr, max_keep, normal)
    assert np.array_equal(keep, rk)
    assert np.array_equal(out.view(np.uint32), ro.view(np.uint32))
    assert (rk[:, 0] == 0).all()                                   # the best box is always kept


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_nms3d_overlap_degenerate(be):
    """Degenerate inputs follow the reference's arithmetic too: identical boxes, zero-height overlap, zero-area boxes
    (0/eps = 0: never suppressed), a box far away; through max_keep = N so that every decision is visible."""
    base = nms3d_boxes(1, 3, 2, seed=5)[0]
    extra = base[:2].copy()                                         # exact duplicates of the two best boxes
    flat = base[2:3].copy(); flat[0, 9] = flat[0, 8]                # zero height
    dot = np.zeros((1, 10), np.float32); dot[0, 9] = 1.0            # zero area at the origin
    above = base[0:1].copy(); above[0, 8] += 50; above[0, 9] += 50  # same footprint, no height overlap
    boxes = np.concatenate([base, extra, flat, dot, dot, above])[None]
    N = boxes.shape[1]
    keep, out = run_nms3d(be, boxes, 0.1, N, False)
    rk, ro = O.nms3d(boxes, 0.1, N, False)
    assert np.array_equal(keep, rk) and np.array_equal(out.view(np.uint32), ro.view(np.uint32))
    L = be.lib
    buf = be.empty(4096)
    assert L.raw("rd_nms3d")(be.ptr(buf), 1, 0, 0.1, 4, 0, be.ptr(buf), be.ptr(buf), be.ptr(buf), 4096, be.stream) == R.RD_ESHAPE
    assert L.raw("rd_nms3d")(be.ptr(buf), 1, 64, 0.1, 4, 0, be.ptr(buf), be.ptr(buf), be.ptr(buf), 16, be.stream) == R.RD_EWORKSPACE


@pytest.mark.parametrize("be", HIP_ONLY, indirect=True)
def test_nms3d_full_size_hip(be):
    """N = 50 000 (rpn_pre_nms_top_n of the shipped configs): same keep list as the lazy greedy oracle, and the keep list
    is a fixed point (no kept pair overlaps above the threshold; every dropped box among the first rows has a kept
    suppressor)."""
    boxes = nms3d_boxes(2, 125, 400, seed=3, jitter=0.15)          # 125 objects x 400 duplicates = 50 000 per frame
    assert boxes.shape[1] == 50000
    keep, out = run_nms3d(be, boxes, 0.1, 500, False)
    rk, ro = O.nms3d(boxes, 0.1, 500, False)
    assert np.array_equal(keep, rk) and np.array_equal(out.view(np.uint32), ro.view(np.uint32))
    for b in range(2):
        k = keep[b][keep[b] >= 0]
        m = O.nms3d_overlap(boxes[b, k], boxes[b, k])
        assert (np.triu(m, 1) <= 0.1).all()
        first = np.setdiff1d(np.arange(2000), k)
        assert (O.nms3d_overlap(boxes[b, k], boxes[b, first]) > 0.1).any(axis=0).all()


def assign_case(N, M, seed):
    """Points around M rotated ground-truth boxes (8 corners: A B C D bottom, E F G H top), the way
    Bbox3dAssigner.get_faster_bbox3d_ind_assigner (rangedet/core/input.py:293-320) calls assign3D_v2."""
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(-40, 40, (M, 2))
    yaw = rng.uniform(-np.pi, np.pi, M)
    l, w, h = rng.uniform(3.5, 9, M), rng.uniform(1.6, 2.6, M), rng.uniform(1.4, 3.0, M)
    z0 = rng.uniform(-1, 0.5, M)
    cor = np.stack([np.stack([l / 2, -w / 2], 1), np.stack([-l / 2, -w / 2], 1), np.stack([-l / 2, w / 2], 1),
                    np.stack([l / 2, w / 2], 1)], 1)                                     # (M,4,2)
    rot = np.stack([np.stack([np.cos(yaw), -np.sin(yaw)], 1), np.stack([np.sin(yaw), np.cos(yaw)], 1)], 1)
    xy = np.einsum('mij,mkj->mki', rot, cor) + ctr[:, None]
    bot = np.concatenate([xy, np.repeat(z0[:, None, None], 4, 1)], 2)
    top = np.concatenate([xy, np.repeat((z0 + h)[:, None, None], 4, 1)], 2)
    gt = np.concatenate([bot, top], 1).astype(np.float32)                               # (M,8,3)
    near = ctr[rng.integers(0, M, N // 2)] + rng.normal(0, 2.5, (N // 2, 2))
    pts = np.concatenate([np.concatenate([near, rng.uniform(-1.5, 4, (N // 2, 1))], 1),
                          np.concatenate([rng.uniform(-75, 75, (N - N // 2, 2)), rng.uniform(-3, 6, (N - N // 2, 1))], 1)])
    pts = pts[rng.permutation(N)].astype(np.float32)
    pts[:M] = gt[:, 0]                                                                  # points exactly on corners / faces
    pts[M:2 * M] = gt.mean(axis=1)
    mask = (rng.uniform(size=N) > 0.1).astype(np.float32)
    nlz = (rng.uniform(size=N) > 0.97).astype(np.float32)
    lim = [float(gt[:, :, 0].max()), float(gt[:, :, 0].min()), float(gt[:, :, 1].max()), float(gt[:, :, 1].min()),
           float(gt[:, :, 2].max()), float(gt[:, :, 2].min())]
    return pts, gt.reshape(M, 24), gt.mean(axis=1), np.full(M, 100, np.float32), mask, nlz, lim


@pytest.mark.parametrize("be", BOTH, indirect=True)
@pytest.mark.parametrize("max_dist", [20.0, 2.0])
def test_assign3d_and_point_num(be, max_dist):
    """rd_assign3d_v2 / rd_get_point_num against the restatement of assigner.h: box index per point exact, counts exact."""
    N = 8000 if be.name == "emu" else 64 * 2650
    pts, gt, ctr, rad, mask, nlz, lim = assign_case(N, 60, seed=4)
    L = be.lib
    out = be.empty(N * 4)
    L.call("rd_assign3d_v2", be.ptr(be.up(pts)), be.ptr(be.up(gt)), be.ptr(be.up(ctr)), be.ptr(be.up(rad)), be.ptr(be.up(mask)),
           be.ptr(be.up(nlz)), N, 60, *lim, max_dist, be.ptr(out), be.stream)
    got = be.down(out, np.int32, (N,))
    ref = O.assign3d_v2(pts, gt, ctr, rad, mask, nlz, *lim, max_dist)
    assert np.array_equal(got, ref)
    assert (ref >= 0).sum() > N // 50 and (ref[(mask < 0.5) | (nlz > 0)] == -1).all()
    nb = L.raw("rd_get_point_num_workspace_bytes")()
    ws, cnt = be.empty(nb), be.empty(N * 4)
    inds = got.astype(np.float32)
    L.call("rd_get_point_num", be.ptr(be.up(inds)), N, be.ptr(cnt), be.ptr(ws), nb, be.stream)
    gc = be.down(cnt, np.float32, (N,))
    assert np.array_equal(gc, O.get_point_num(inds))
    hist = np.bincount(got[got >= 0], minlength=60)
    assert np.array_equal(gc[got >= 0], hist[got[got >= 0]].astype(np.float32)) and (gc[got < 0] == -1).all()
    buf = be.empty(4096)
    assert L.raw("rd_assign3d_v2")(*([be.ptr(buf)] * 6), 10, 0, *lim, 20.0, be.ptr(buf), be.stream) == R.RD_ESHAPE
    assert L.raw("rd_assign3d_v2")(*([be.ptr(buf)] * 6), 10, 5000, *lim, 20.0, be.ptr(buf), be.stream) == R.RD_ESHAPE


@pytest.mark.parametrize("be", HIP_ONLY, indirect=True)
def test_processing_cxx_module_surface(be):
    """The drop-in module, called the way the reference's callers do (tools/test.py:211-218, rangedet/core/input.py:311-319,
    433-435): same positional arguments, same result shapes."""
    from rangedet_amd import processing_cxx
    d = synth.cluster_dets(12, 8, seed=2)
    flat, keep = processing_cxx.wnms_4c(d, 0.1, 0.5, False, 100)
    rflat, rkeep = O.wnms_4c(d, 0.1, 0.5, False, 100)
    assert keep == rkeep and np.array_equal(np.array(flat, np.float32).view(np.uint32), np.array(rflat, np.float32).view(np.uint32))
    assert processing_cxx.wnms_4c(np.zeros((0, 12), np.float32), 0.1, 0.5, False, 100) == ([], [])
    pts, gt, ctr, rad, mask, nlz, lim = assign_case(4096, 20, seed=9)
    inds = processing_cxx.assign3D_v2(pts.reshape(-1, 3), gt.reshape(-1, 24), ctr.reshape(-1, 3), rad.reshape(-1, 1),
                                      mask.reshape(-1, 1), nlz.reshape(-1, 1), *lim, 20.0)
    assert inds.shape == (4096, 1) and inds.dtype == np.int32
    assert np.array_equal(inds.reshape(-1), O.assign3d_v2(pts, gt, ctr, rad, mask, nlz, *lim, 20.0))
    num = processing_cxx.get_point_num(inds.reshape(-1).astype(np.float32))
    assert num.shape == (4096, 1) and np.array_equal(num.reshape(-1), O.get_point_num(inds.astype(np.float32)))
    w = 1 / num.reshape(-1)                                        # get_normalization_weight, input.py:436-437
    w[w == -1] = 0
    assert np.isfinite(w).all()


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_error_conventions(be):
    """Status codes instead of aborts; messages through rd_last_error_string (SURVEY.md 8b 'Error conventions')."""
    L = be.lib
    buf = be.empty(4096)
    p = be.ptr(buf)
    rc = L.raw("rd_conv2d_bn_act")(p, 16, 0, p, p, p, None, 0, 0, p, 96, 0, 1, 2, 8, 16, 96, 3, 3, 1, 0, F32, be.stream)
    assert rc == R.RD_ESHAPE and b"cout" in L.raw("rd_last_error_string")()
    rc = L.raw("rd_conv2d_bn_act")(p, 16, 0, p, p, p, None, 0, 0, p, 64, 0, 1, 2, 8, 16, 64, 5, 5, 1, 0, F32, be.stream)
    assert rc == R.RD_ESHAPE
    rc = L.raw("rd_wnms_4c")(p, 100000, None, None, 0, 0.1, 0.5, 0, 100, p, p, p, p, 4096, be.stream)
    assert rc == R.RD_ESHAPE
    rc = L.raw("rd_wnms_4c")(p, 64, None, None, 0, 0.1, 0.5, 0, 100, p, p, p, p, 16, be.stream)
    assert rc == R.RD_EWORKSPACE
    rc = L.raw("rd_decode3d_bbox")(None, p, p, 1, 10, 8, 0, be.stream)
    assert rc == R.RD_EINVAL
    assert L.raw("rd_version")() >= 100


@pytest.mark.gpu
def test_rotated_box_kernels_unaffected_by_concurrent_convs():
    """Rotated IoU ('bev' and '3d') and NMS3D are cross-product code like the weighted NMS, and the round-4 build had the swapped packed-fp32
    form in them too (wrong next to MFMA waves).  Each op's result on an idle GPU must come back bit for bit while
    64->64 convolutions run on a second stream.  (The WNMS chain has its own test of this kind in tests/test_dist.py.)"""
    L = R.get_lib()
    dev = "cuda"
    cur = torch.cuda.current_stream().cuda_stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    B, N, c, H, W = 2, 60000, 64, 64, 2656
    prop, gt = _riou_case(B, N, 40, seed=11)
    dp, dg = torch.from_numpy(prop).to(dev), torch.from_numpy(gt).to(dev)
    g7 = torch.from_numpy(np.stack([_gt7_from_corners(gt[b], -1.2, 0.6) for b in range(B)])).to(dev)          # (B, 200, 7)
    boxes = torch.from_numpy(nms3d_boxes(4, 60, 40, seed=3)).to(dev)                       # (4, 2400, 10)
    nb, nn = boxes.shape[:2]
    wb = L.raw("rd_nms3d_workspace_bytes")(nn, nb)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    outs = dict(bev=torch.zeros(B, N, device=dev), bev_arg=torch.zeros(B, N, dtype=torch.int32, device=dev), i3d=torch.zeros(B, N, device=dev),
                keep=torch.zeros(nb, 500, dtype=torch.int32, device=dev), rows=torch.zeros(nb, 500, 10, device=dev))

    def ops(st):
        L.call("rd_batch_rotated_iou", dp.data_ptr(), 10, dg.data_ptr(), outs["bev"].data_ptr(), outs["bev_arg"].data_ptr(), B, N, 200, st)
        L.call("rd_batch_rotated_iou_3d", dp.data_ptr(), 10, g7.data_ptr(), outs["i3d"].data_ptr(), None, B, N, 200, st)
        L.call("rd_nms3d", boxes.data_ptr(), nb, nn, 0.3, 500, 0, outs["keep"].data_ptr(), outs["rows"].data_ptr(), ws.data_ptr(), wb, st)

    ops(cur)
    torch.cuda.synchronize()
    base = {k: v.clone() for k, v in outs.items()}
    assert float(base["bev"].max()) > 0.5 and float(base["i3d"].max()) > 0.3 and int((base["keep"] >= 0).sum()) > 100
    x = torch.randn(8 * H * W * c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.from_numpy(L.pack_conv3x3_ex(np.random.RandomState(0).randn(c, c, 3, 3).astype(np.float32) * 0.05, 1, c, fold_scale=np.ones(c, np.float32),
                                           dtype=R.RD_BF16)).to(dev)
    sh = torch.zeros(c, device=dev)
    bad = {k: 0 for k in outs}
    for _ in range(12):
        for v in outs.values():
            v.zero_()
        torch.cuda.synchronize()
        for _ in range(8):
            L.call("rd_conv3x3_bn_act_ex", x.data_ptr(), c, 0, w.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                   y.data_ptr(), c, 0, 8, H, W, c, c, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, R.RD_BF16, s1.cuda_stream)
        ops(s2.cuda_stream)
        torch.cuda.synchronize()
        for k in outs:
            bad[k] += not torch.equal(outs[k].view(torch.int32), base[k].view(torch.int32))
    assert not any(bad.values()), "replays next to the convolutions that differ from the idle result, of 12: %r" % bad
