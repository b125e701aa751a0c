"""Single-operator test hooks of the C-ABI (``spk_op_*`` in include/sykepic_hip.h) on torch device tensors.

Each call runs exactly the launches ``spk_train_forward_backward`` makes for ONE Conv2d + BatchNorm2d layer
(the reference reaches them through ``net(x)`` / ``loss.backward()``, sykepic/train/train.py:240,242), so a
parity test can feed a kernel known operands.  Tensor conventions: activations and gradients are torch
``bfloat16`` tensors in NCHW *logical* shape; they are handed to the library as NHWC (``channels_last``
storage), weights as float32 OIHW (handed over as [Cout][kh][kw][Cin]).  No CPU fallback.
"""

import ctypes as C

import torch

from . import lib


def _nhwc(t):
    """NCHW logical tensor -> contiguous [N,H,W,C] bf16 device tensor."""
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t_nhwc):
    return t_nhwc.permute(0, 3, 1, 2)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _is_stem(cin, cout, k, stride, pad):
    return cin <= 4 and k == 7 and stride == 2 and pad == 3 and cout == 64


def _stem_input(x):
    """[N,C<=4,H,W] -> NHWC with 4 stored channels and an even width (the layout spk_launch_to_nhwc4 produces)."""
    n, c, h, w = x.shape
    wp = (w + 1) & ~1
    out = torch.zeros((n, h, wp, 4), dtype=torch.bfloat16, device=x.device)
    out[:, :, :w, :c] = x.permute(0, 2, 3, 1)
    return out


def conv_bn_train_forward(x, weight, gamma, beta, running_mean, running_var, res=None, relu=True, stride=1, pad=0):
    """Returns dict(out, raw, mask, mean, invstd); running_mean / running_var (float32 device tensors) are updated
    in place."""
    so = lib.load()
    dev = x.device
    n, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    stem = _is_stem(cin, cout, k, stride, pad)
    xh = _stem_input(x) if stem else _nhwc(x.to(torch.bfloat16))
    wk = weight.float().permute(0, 2, 3, 1).contiguous()
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((n, oh, ow, cout), dtype=torch.bfloat16, device=dev)
    raw = torch.empty_like(out)
    mask = torch.zeros((n * oh * ow, cout // 8), dtype=torch.uint8, device=dev)
    stats = torch.empty((2, cout), dtype=torch.float32, device=dev)
    resh = _nhwc(res.to(torch.bfloat16)) if res is not None else None
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv_bn_train_forward(
            _p(xh), _p(wk), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(resh), _p(out), _p(raw),
            _p(mask), _p(stats), n, h, w, cin, cout, k, stride, pad, int(bool(relu)), _stream(dev)))
    return {"out": _nchw(out), "raw": _nchw(raw), "mask": mask, "mean": stats[0], "invstd": stats[1]}


def bn_backward(g, mask, raw, mean, invstd, gamma, relu=True, g_res=None, res_accumulate=False, want_res=False):
    """g, raw: [N,C,H,W] bf16.  Returns dict(dy, dgamma, dbeta, g_res)."""
    so = lib.load()
    dev = g.device
    n, c, h, w = g.shape
    gh, rh = _nhwc(g), _nhwc(raw)
    dy = torch.empty_like(gh)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    gres = None
    if want_res or g_res is not None:
        gres = _nhwc(g_res.to(torch.bfloat16)) if g_res is not None else torch.empty_like(gh)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_bn_backward(_p(gh), _p(mask), _p(rh), _p(mean), _p(invstd), _p(gamma), _p(dgamma),
                                        _p(dbeta), _p(dy), _p(gres), int(bool(res_accumulate)), n * h * w, c,
                                        int(bool(relu)), _stream(dev)))
    return {"dy": _nchw(dy), "dgamma": dgamma, "dbeta": dbeta, "g_res": _nchw(gres) if gres is not None else None}


def conv_dgrad(dy, weight, in_hw, stride=1, pad=0, accumulate_into=None):
    """dx [N,Cin,H,W] bf16 = conv_transpose(dy, weight) (+ accumulate_into)."""
    so = lib.load()
    dev = dy.device
    n, cout = dy.shape[:2]
    _, cin, k, _ = weight.shape
    h, w = in_hw
    dyh = _nhwc(dy.to(torch.bfloat16))
    wk = weight.float().permute(0, 2, 3, 1).contiguous()
    if accumulate_into is not None:
        dx = _nhwc(accumulate_into.to(torch.bfloat16))
    else:
        # poison: every element must be written (or zeroed) by the launches
        dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv_dgrad(_p(dyh), _p(wk), _p(dx), int(accumulate_into is not None), n, h, w, cin, cout,
                                       k, stride, pad, _stream(dev)))
    return _nchw(dx)


def conv_dgrad_bn_backward(dy, weight, in_hw, raw, mask, mean, invstd, gamma, pad=0, relu=True, accumulate_into=None,
                           res_src=None, res_bits=None):
    """Stride-1 data gradient whose epilogue also makes the BatchNorm-backward sums of the layer that produced the conv's
    input, then that layer's finalize + apply (spk_op_conv_dgrad_bn_backward).  raw [N,Cin,H,W] bf16, mask as returned by
    `conv_bn_train_forward`.  res_src [N,Cin,H,W] bf16 + res_bits (instead of accumulate_into): the shortcut gradient
    res_src * bit added at its source.  Returns dict(g, dy, dgamma, dbeta): g = the stored input gradient."""
    so = lib.load()
    dev = dy.device
    n, cout = dy.shape[:2]
    _, cin, k, _ = weight.shape
    h, w = in_hw
    dyh = _nhwc(dy.to(torch.bfloat16))
    wk = weight.float().permute(0, 2, 3, 1).contiguous()
    if accumulate_into is not None:
        dx = _nhwc(accumulate_into.to(torch.bfloat16))
    else:
        dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    rawh = _nhwc(raw.to(torch.bfloat16))
    rsh = _nhwc(res_src.to(torch.bfloat16)) if res_src is not None else None
    dyp = torch.empty_like(rawh)
    dgamma = torch.empty(cin, dtype=torch.float32, device=dev)
    dbeta = torch.empty(cin, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv_dgrad_bn_backward(
            _p(dyh), _p(wk), _p(dx), int(accumulate_into is not None), _p(rawh), _p(mask), _p(mean), _p(invstd), _p(gamma),
            _p(dgamma), _p(dbeta), _p(dyp), n, h, w, cin, cout, k, pad, int(bool(relu)), _p(rsh), _p(res_bits), _stream(dev)))
    return {"g": _nchw(dx), "dy": _nchw(dyp), "dgamma": dgamma, "dbeta": dbeta}


def conv_wgrad(x, dy, k, stride=1, pad=0):
    """dw [Cout,Cin,k,k] float32 from x [N,Cin,H,W] and dy [N,Cout,Ho,Wo] (bf16)."""
    so = lib.load()
    dev = x.device
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    stem = _is_stem(cin, cout, k, stride, pad)
    xh = _stem_input(x.to(torch.bfloat16)) if stem else _nhwc(x.to(torch.bfloat16))
    dyh = _nhwc(dy.to(torch.bfloat16))
    dw = torch.full((cout, k, k, cin), float("nan"), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv_wgrad(_p(xh), _p(dyh), _p(dw), n, h, w, cin, cout, k, stride, pad, _stream(dev)))
    return dw.permute(0, 3, 1, 2)


def pw_fp8(x, weight, bn_scale, bn_bias, act=0, a_scale=1.0, y_scale=1.0, out_fp8=False, res=None, gate=None, hw=1):
    """fp8 pointwise conv (spk_op_pw_fp8).  x: [M, cin] float16, or uint8 holding e4m3 bytes (value = byte *
    a_scale); weight [cout, cin] float32; returns [M, cout] float16, or uint8 e4m3 bytes when out_fp8."""
    so = lib.load()
    dev = x.device
    m, cin = x.shape
    cout = weight.shape[0]
    a_fp8 = x.dtype == torch.uint8
    x = x.contiguous()
    y = torch.empty((m, cout), dtype=torch.uint8 if out_fp8 else torch.float16, device=dev)
    w = weight.float().contiguous()
    with torch.cuda.device(dev):
        lib.check(so.spk_op_pw_fp8(_p(x), int(a_fp8), _p(w), _p(y), int(bool(out_fp8)),
                                   _p(res.contiguous()) if res is not None else None, _p(bn_scale.float().contiguous()),
                                   _p(bn_bias.float().contiguous()),
                                   _p(gate.float().contiguous()) if gate is not None else None, int(hw), m, cin, cout,
                                   int(act), float(a_scale), float(y_scale), _stream(dev)))
    return y


def dwconv(x, weight, bn_scale, bn_bias, k, stride=1, act=2, lds=1):
    """Depthwise conv + folded BN + activation (spk_op_dwconv).  x [N,C,H,W] float16, weight [C,1,k,k];
    returns (y [N,C,Ho,Wo] float16, pool [N,C] float32 sums of y before rounding)."""
    so = lib.load()
    dev = x.device
    n, c, h, w = x.shape
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    y = torch.full((n, ho, wo, c), float("nan"), dtype=torch.float16, device=dev)
    pool = torch.empty((n, c), dtype=torch.float32, device=dev)
    wk = weight.float().reshape(c, k * k).contiguous()
    with torch.cuda.device(dev):
        lib.check(so.spk_op_dwconv(_p(xh), _p(wk), _p(bn_scale.float().contiguous()), _p(bn_bias.float().contiguous()),
                                   _p(y), _p(pool), n, h, w, c, k, stride, int(act), int(lds), _stream(dev)))
    return y.permute(0, 3, 1, 2), pool


def conv1x1_num_configs():
    return int(lib.load().spk_op_conv1x1_num_configs())


def conv1x1(x, weight, bn_scale, bn_bias, stride=1, relu=True, res=None, split=True, cfg=0):
    """Eval-path 1x1 conv + folded BN (+ shortcut) (+ ReLU) (spk_op_conv1x1).  x [N,Cin,H,W] float16, weight
    [Cout,Cin] (or [Cout,Cin,1,1]); res [N,Cout,Ho,Wo] float16 or None.  cfg >= 0: that configuration of the
    direct-operand kernel (raises RuntimeError "does not fit" when it cannot run the problem), cfg < 0: the implicit
    GEMM.  Returns [N,Cout,Ho,Wo] float16."""
    so = lib.load()
    dev = x.device
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    rh = res.half().permute(0, 2, 3, 1).contiguous() if res is not None else None
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.float16, device=dev)
    wk = weight.float().reshape(cout, cin).contiguous()
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv1x1(_p(xh), _p(wk), _p(bn_scale.float().contiguous()), _p(bn_bias.float().contiguous()),
                                    _p(rh) if rh is not None else None, _p(y), n, h, w, cin, cout, int(stride),
                                    int(bool(relu)), int(bool(split)), int(cfg), _stream(dev)))
    return y.permute(0, 3, 1, 2)


def conv1x1_chain(x, weight, bn_scale, bn_bias, res, weight_z, bnz_scale, bnz_bias, relu=True, relu_z=True):
    """Two chained 1x1 convs in one launch (spk_op_conv1x1_chain): returns (y [N,256,H,W], z [N,Coutz,H,W]) float16."""
    so = lib.load()
    dev = x.device
    n, cin, h, w = x.shape
    cout, coutz = weight.shape[0], weight_z.shape[0]
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    rh = res.half().permute(0, 2, 3, 1).contiguous()
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.float16, device=dev)
    z = torch.full((n, h, w, coutz), float("nan"), dtype=torch.float16, device=dev)
    f = lambda t: t.float().contiguous()   # noqa: E731
    wk, wzk = f(weight.reshape(cout, cin)), f(weight_z.reshape(coutz, cout))
    s1, b1, s2, b2 = f(bn_scale), f(bn_bias), f(bnz_scale), f(bnz_bias)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv1x1_chain(_p(xh), _p(wk), _p(s1), _p(b1), _p(rh), _p(y), _p(wzk), _p(s2), _p(b2), _p(z), n, h, w,
                                          cin, cout, coutz, int(bool(relu)), int(bool(relu_z)), _stream(dev)))
    return y.permute(0, 3, 1, 2), z.permute(0, 3, 1, 2)


def conv1x1_dual(x, w1, bn1_scale, bn1_bias, x2, w2, bn2_scale, bn2_bias, stride2=2, relu=True, split=False, cfg=0):
    """A block-closing 1x1 conv and the block's 1x1 shortcut conv as one K-concatenated GEMM (spk_op_conv1x1_dual):
    act(BN1(W1 . x) + BN2(W2 . x2[::stride2, ::stride2])).  x [N,Cin,Ho,Wo], x2 [N,Cin2,H2,W2] float16; returns
    [N,Cout,Ho,Wo] float16.  Raises RuntimeError "does not fit" when `cfg` cannot run the problem."""
    so = lib.load()
    dev = x.device
    n, cin, ho, wo = x.shape
    _, cin2, h2, w2d = x2.shape
    cout = w1.shape[0]
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    x2h = x2.half().permute(0, 2, 3, 1).contiguous()
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.float16, device=dev)
    f = lambda t: t.float().contiguous()   # noqa: E731
    w1k, w2k = f(w1.reshape(cout, cin)), f(w2.reshape(cout, cin2))
    s1, b1, s2, b2 = f(bn1_scale), f(bn1_bias), f(bn2_scale), f(bn2_bias)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv1x1_dual(_p(xh), _p(w1k), _p(s1), _p(b1), _p(x2h), _p(w2k), _p(s2), _p(b2), _p(y), n, ho, wo,
                                         cin, h2, w2d, cin2, cout, int(stride2), int(bool(relu)), int(bool(split)), int(cfg),
                                         _stream(dev)))
    return y.permute(0, 3, 1, 2)


def conv3x3_num_configs():
    return int(lib.load().spk_op_conv3x3_num_configs())


def conv3x3(x, weight, bn_scale, bn_bias, relu=True, split=False, cfg=0, res=None):
    """Eval-path 3x3 stride-1 pad-1 conv + folded BN (+ shortcut) (+ ReLU) (spk_op_conv3x3).  x [N,Cin,H,W] float16,
    weight [Cout,Cin,3,3], res [N,Cout,H,W] or None; cfg >= 0: that configuration of the LDS-window kernel, cfg < 0: the
    implicit GEMM."""
    so = lib.load()
    dev = x.device
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.float16, device=dev)
    wk = weight.float().permute(0, 2, 3, 1).contiguous()   # OIHW -> O,kh,kw,I
    rh = res.half().permute(0, 2, 3, 1).contiguous() if res is not None else None
    with torch.cuda.device(dev):
        lib.check(so.spk_op_conv3x3(_p(xh), _p(wk), _p(bn_scale.float().contiguous()), _p(bn_bias.float().contiguous()),
                                    _p(rh) if rh is not None else None, _p(y), n, h, w, cin, cout, int(bool(relu)), int(bool(split)), int(cfg),
                                    _stream(dev)))
    return y.permute(0, 3, 1, 2)


def bottleneck(x, w1, w2, w3, bn1, bn2, bn3, fused=True, iters=0, stamps=None, wz=None, bnz=None):
    """Eval-path identity bottleneck block (spk_op_bottleneck): x [N,4cm,H,W] float16, w1 [cm,4cm,1,1], w2 [cm,cm,3,3],
    w3 [4cm,cm,1,1], bn* = (scale, shift) of the folded eval BatchNorms.  fused: the one-kernel form (csrc/conv_bneck.hip),
    else the eval path's three launches.  Returns y [N,4cm,H,W] (and the mean milliseconds per block when iters > 0)."""
    import ctypes as C
    so = lib.load()
    dev = x.device
    n, c4, h, w = x.shape
    cm = w1.shape[0]
    xh = x.half().permute(0, 2, 3, 1).contiguous()
    y = torch.full((n, h, w, c4), float("nan"), dtype=torch.float16, device=dev)
    k1 = w1.float().reshape(cm, c4).contiguous()
    k2 = w2.float().permute(0, 2, 3, 1).contiguous()
    k3 = w3.float().reshape(c4, cm).contiguous()
    vecs = [t.float().contiguous() for pair in (bn1, bn2, bn3) for t in pair]
    ms = C.c_float(0.0)
    # fused: False / 0 three launches, True / 1 the whole block as one kernel, 2 conv1 + (conv2 + conv3 + shortcut [+ wz]) kernel
    kz = z = sz = bz = None
    coutz = 0
    if wz is not None:      # the conv + BN + ReLU that reads the block's output (the next block's conv1): z comes back too
        coutz = wz.shape[0]
        kz = wz.float().reshape(coutz, c4).contiguous()
        sz, bz = (t.float().contiguous() for t in bnz)
        z = torch.full((n, h, w, coutz), float("nan"), dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        lib.check(so.spk_op_bottleneck(_p(xh), _p(k1), _p(k2), _p(k3), *[_p(v) for v in vecs], _p(y), n, h, w, cm,
                                       int(fused), int(iters), C.cast(C.pointer(ms), C.c_void_p), _stream(dev),
                                       _p(stamps) if stamps is not None else None,
                                       _p(kz) if kz is not None else None, _p(sz) if sz is not None else None,
                                       _p(bz) if bz is not None else None, _p(z) if z is not None else None, coutz))
    out = y.permute(0, 3, 1, 2)
    if z is not None:
        out = (out, z.permute(0, 3, 1, 2))
    return (out, float(ms.value)) if iters > 0 else out


def zero_sum_round(w, mu=None, period=None):
    """w [rows, row_len] float32 device tensor, mu [period] or None -> the zero-sum rounded rows (float32 values that are
    fp16 numbers), csrc/zero_sum.hip."""
    so = lib.load()
    w = w.float().contiguous()
    out = torch.empty_like(w)
    rows, row_len = w.shape
    if mu is not None:
        mu = mu.float().contiguous()
        period = period or mu.numel()
    with torch.cuda.device(w.device):
        lib.check(so.spk_op_zero_sum_round(_p(w), _p(mu), _p(out), rows, row_len, int(period or row_len), _stream(w.device)))
    return out
