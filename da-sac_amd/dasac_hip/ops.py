"""Tensor-level wrappers over the C ABI (one function per kernel family)."""
import os

import torch

from . import lib as L


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Profile:
    """Optional per-launch HIP-event timing of the GEMM kernels (bench.py's roofline leg).  Events are
    recorded on the stream the kernels are launched on (torch's current stream); off by default."""

    def __init__(self):
        self.on, self.records = False, []

    def start(self):
        self.on, self.records = True, []

    def stop(self, by_shape=False):
        """Per-kernel-class totals; by_shape=True keys on (class, shape tag) instead (tools/step_shapes.py)."""
        self.on = False
        torch.cuda.synchronize()
        out = {}
        for name, flops, a, b, tag, nbytes in self.records:
            d = out.setdefault((name, tag) if by_shape else name, {"flops": 0.0, "bytes": 0.0, "seconds": 0.0, "launches": 0})
            d["flops"] += flops
            d["bytes"] += nbytes
            d["seconds"] += a.elapsed_time(b) * 1e-3
            d["launches"] += 1
        self.records = []
        return out

    def span(self, name, flops, tag=None, nbytes=0.0):
        """flops / nbytes: ALGORITHMIC work of the launch (2*M*N*K; every operand once)."""
        return _Span(self, name, flops, tag, nbytes) if self.on else _NULL


class _Span:
    def __init__(self, prof, name, flops, tag, nbytes=0.0):
        self.prof, self.name, self.flops, self.tag, self.nbytes = prof, name, flops, tag, nbytes

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        self.prof.records.append((self.name, self.flops, self.a, b, self.tag, self.nbytes))


class _Null:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


_NULL = _Null()
PROFILE = _Profile()


def pseudo_labels(probs, ignore, upper, lower, disc=None, want_idx=False):
    """models/sac.py:154-187.  probs [B,C,H,W] f32 cuda; ignore bool [B,H,W] or None; disc [C] or None.
    Returns (labels i64 [B,H,W], max_conf f32 [B,1,H,W], max_idx i64 [B,1,H,W] or None)."""
    L.require_gpu(probs, ignore, disc)
    lib = L.load()
    probs = _c(probs)
    B, Cn, H, W = probs.shape
    HW = H * W
    labels = torch.empty((B, H, W), dtype=torch.int64, device=probs.device)
    conf = torch.empty((B, 1, H, W), dtype=torch.float32, device=probs.device)
    idx = torch.empty((B, 1, H, W), dtype=torch.int64, device=probs.device) if want_idx else None
    ign = None if ignore is None else _c(ignore).view(torch.uint8)
    ws_bytes = lib.dasac_pseudo_labels_workspace(B, Cn, HW)
    ws = L.workspace(ws_bytes, probs.device)
    with PROFILE.span("pseudo_labels", 0.0, None, probs.numel() * 4.0 + B * HW * (1 + 8 + 4 + (8 if want_idx else 0))):
        L.check(lib.dasac_pseudo_labels(probs.data_ptr(), L.ptr(ign), L.ptr(disc), float(upper), float(lower), B, Cn, HW,
                                        labels.data_ptr(), conf.data_ptr(), L.ptr(idx), ws.data_ptr(), ws.numel(),
                                        L.stream_ptr()), "dasac_pseudo_labels")
    return labels, conf, idx


# ----------------------------------------------------------------------------------------------
# convolution as implicit GEMM (include/dasac_hip.h: dasac_conv_*)
# ----------------------------------------------------------------------------------------------
class ConvSpec:
    """Static description of one (possibly multi-branch) convolution.
    branches: list of (kh, kw, dilation, padding); stride applies to all."""

    def __init__(self, cin, cout, branches, stride=1):
        self.cin, self.cout, self.stride = int(cin), int(cout), int(stride)
        self.branches = [tuple(int(v) for v in b) for b in branches]
        self.taps = sum(b[0] * b[1] for b in self.branches)
        self.K = self.taps * self.cin            # forward / wgrad contraction length
        self.Kt = self.taps * self.cout          # data-gradient contraction length

    def out_hw(self, h, w):
        kh, kw, d, p = self.branches[0]
        oh = (h + 2 * p - d * (kh - 1) - 1) // self.stride + 1
        ow = (w + 2 * p - d * (kw - 1) - 1) // self.stride + 1
        return oh, ow


_i32 = torch.int32

# Arithmetic of the forward / data-gradient GEMMs:
#   "fp32"   exact fp32 products on v_mfma_f32_32x32x2_f32 (default);
#   "bf16x3" every operand split into bf16 head + tail, three v_mfma_f32_32x32x16_bf16 per product, fp32
#            accumulation (dasac_conv_gemm_x3; ~2^-16 relative error per product, inside the 1e-3 parity bar).
PRECISION = os.environ.get("DASAC_PRECISION", "fp32")


def set_precision(mode):
    global PRECISION
    assert mode in ("fp32", "bf16x3"), mode
    PRECISION = mode


def _finish_pack(packed, M, K):
    """Converts a freshly packed fp32 operand to the split-bf16 layout (in place, same size) when selected."""
    lib = L.load()
    x3 = PRECISION == "bf16x3" and lib.dasac_conv_mpad(M) >= 64
    if x3:
        L.check(lib.dasac_conv_pack_x3(packed.data_ptr(), M, K, packed.data_ptr(), L.stream_ptr()), "dasac_conv_pack_x3")
    packed.dasac_x3 = x3
    return packed


def gemm_order(spec, transposed):
    """K order of the forward / data-gradient GEMM: chunk-major (1) for multi-tap convs whose gathered channel
    count is a multiple of 16, tap-major (0) otherwise.  The weight gradient always uses tap-major tables."""
    C = spec.cout if transposed else spec.cin
    return 1 if (spec.taps > 1 and C % 16 == 0 and os.environ.get("DASAC_KORDER", "1") != "0") else 0


def conv_table(spec, plane_h, plane_w, transposed, device, order=0):
    """Gather table for planes of plane_h x plane_w (forward: the input planes; transposed: dz planes)."""
    lib = L.load()
    C = spec.cout if transposed else spec.cin
    K = spec.taps * C
    table = torch.empty((lib.dasac_conv_kpad(K), 4), dtype=_i32, device=device)
    cols = [torch.tensor(c, dtype=_i32) for c in zip(*spec.branches)]   # host arrays: kh, kw, dil, pad
    L.check(lib.dasac_conv_table(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), cols[3].data_ptr(),
                                 len(spec.branches), C, plane_h, plane_w, int(transposed), int(order), table.data_ptr(),
                                 L.stream_ptr()), "dasac_conv_table")
    return table


def conv_pack(spec, weights, transposed, scale=None, out=None, order=0):
    """Packs the branch weight tensors [Cout,Cin,kh,kw] into the [Kpad][Mpad] GEMM operand."""
    lib = L.load()
    L.require_gpu(*weights)
    M = spec.cin if transposed else spec.cout
    K = spec.Kt if transposed else spec.K
    shape = (lib.dasac_conv_kpad(K), lib.dasac_conv_mpad(M))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=weights[0].device)
    tap0 = 0
    for w, (kh, kw, _, _) in zip(weights, spec.branches):
        L.check(lib.dasac_conv_pack(_c(w).data_ptr(), L.ptr(scale), spec.cout, spec.cin, kh * kw, tap0, spec.taps,
                                    int(transposed), int(order), out.data_ptr(), L.stream_ptr()), "dasac_conv_pack")
        tap0 += kh * kw
    return _finish_pack(out, M, K)


RELU_BITS = os.environ.get("DASAC_RELU_BITS", "1") != "0"      # ReLU patterns as bit masks (1/32 of the bytes) between fwd and dgrad


def bits_ok(M, Cx):
    """True when the fp32 conv GEMM for an output of M channels over Cx gathered channels has a bit-mask variant."""
    return RELU_BITS and PRECISION == "fp32" and bool(L.load().dasac_conv_gemm_bits_ok(int(M), int(Cx)))


class ReluBits:
    """ReLU pattern of one activation tensor [Nb, M, OH, OW] as bits (include/dasac_hip.h: dasac_conv_gemm relu_bits_out)."""

    def __init__(self, Nb, M, OH, OW, device):
        self.shape = (Nb, M, OH, OW)
        self.words = torch.empty(L.load().dasac_relu_bits_words(M, Nb * OH * OW), dtype=torch.int32, device=device)


def stats_ok(M, Cx):
    """True when the conv GEMM for M output channels over Cx gathered channels can leave per-tile channel statistics."""
    return PRECISION == "fp32" and os.environ.get("DASAC_GEMM_STATS", "1") != "0" and bool(L.load().dasac_conv_gemm_stats_ok(int(M), int(Cx)))


def tile_stats_buffer(Nb, M, OH, OW, device):
    """[pixel tiles, 2, Mpad] fp32: filled by conv_gemm(stats=...) (every slot written), read by bn_train_forward."""
    lib = L.load()
    return torch.empty((lib.dasac_conv_gemm_stats_tiles(Nb, OH, OW), 2, lib.dasac_conv_mpad(M)), dtype=torch.float32, device=device)


def conv_gemm(x, packed, table, out, grid_hw, stride, M, K, ostride=1, shift=None, res=None, mask=None, relu=False, bits_out=None,
              stats=None, schedule=None):
    """out[n,m,oh*os,ow*os] = epilogue(sum_k packed[k][m] * gather(x)); see dasac_conv_gemm.
    mask: fp32 activation (zero where <= 0) or a ReluBits of the output's shape; bits_out: ReluBits to fill (relu only);
    stats: `tile_stats_buffer` to fill with per-tile channel sums / sums of squares of the output (dasac_conv_gemm_stats).
    schedule: None = the library's choice (one block per tile; long-K layers with a ragged last round: whole rounds one block per
    tile + a split-K tail in the same launch; few tiles x long K: the persistent stream-K kernel); 1 / 2 force one block per tile /
    the persistent stream-K kernel (tests, tools)."""
    lib = L.load()
    L.require_gpu(x, packed, table, out)
    Nb, Cx, H, W = x.shape
    OH, OW = grid_hw
    assert out.shape[0] == Nb and out.shape[1] == M and x.is_contiguous() and out.is_contiguous()
    mask_bits = None
    if isinstance(mask, ReluBits):
        assert mask.shape == tuple(out.shape) and ostride == 1
        mask_bits, mask = mask.words, None
    if bits_out is not None:
        assert relu and ostride == 1 and bits_out.shape == tuple(out.shape)
    for t_ in (res, mask):
        assert t_ is None or (t_.shape == out.shape and t_.is_contiguous())
    ws = L.workspace(lib.dasac_conv_gemm_workspace(), x.device, owner="conv_gemm")
    fn = lib.dasac_conv_gemm_x3 if getattr(packed, "dasac_x3", False) else lib.dasac_conv_gemm
    tag = (M, K, Nb * OH * OW, stride, ostride, res is not None, mask is not None or mask_bits is not None)

    def launch(span, pix_begin, pix_count, schedule):
        n = pix_count if pix_count else Nb * OH * OW - pix_begin
        frac = n / float(Nb * OH * OW)
        nbytes = 4.0 * (x.numel() * frac + packed.numel() + n * M * (1 + (res is not None) + (mask is not None)
                                                                     + ((mask_bits is not None) + (bits_out is not None)) / 32.0))
        if stats is not None:
            assert mask is None and mask_bits is None and bits_out is None and stats.is_contiguous() and stats.dtype == torch.float32
            with PROFILE.span(span, 2.0 * n * M * K, tag, nbytes):
                L.check(lib.dasac_conv_gemm_stats(x.data_ptr(), packed.data_ptr(), table.data_ptr(), out.data_ptr(), Nb, Cx, H, W, OH, OW,
                                                  stride, M, K, out.shape[2], out.shape[3], ostride, L.ptr(shift), L.ptr(res), int(relu),
                                                  pix_begin, pix_count, schedule, L.ptr(ws), 0 if ws is None else ws.numel(),
                                                  stats.data_ptr(), L.stream_ptr()), "dasac_conv_gemm_stats")
            return
        with PROFILE.span(span, 2.0 * n * M * K, tag, nbytes):
            L.check(fn(x.data_ptr(), packed.data_ptr(), table.data_ptr(), out.data_ptr(), Nb, Cx, H, W, OH, OW,
                       stride, M, K, out.shape[2], out.shape[3], ostride, L.ptr(shift), L.ptr(res),
                       L.ptr(mask), L.ptr(mask_bits), 0 if bits_out is None else bits_out.words.data_ptr(), int(relu),
                       pix_begin, pix_count, schedule, L.ptr(ws), 0 if ws is None else ws.numel(),
                       L.stream_ptr()), "dasac_conv_gemm")

    if schedule is not None:
        launch("conv_gemm<stream-K>" if schedule == 2 else "conv_gemm<tile-per-block>", 0, 0, int(schedule))
        return out
    if lib.dasac_conv_gemm_tail_split(Nb, OH, OW, M, K) > 0:
        # ONE launch: whole rounds one block per tile (lockstep over K: halo rows shared in L2) + the remaining tiles cut into
        # K-ranges that fill the chip once more (round 6; rounds 2-5 issued that remainder as a second, persistent stream-K launch)
        launch("conv_gemm<tile+tail>", 0, 0, 0)
        return out
    sk = PROFILE.on and lib.dasac_conv_gemm_schedule(Nb, OH, OW, M, K)
    launch("conv_gemm<stream-K>" if sk else "conv_gemm<tile-per-block>", 0, 0, 0)
    return out


def conv_forward(spec, x, weights, scale=None, shift=None, res=None, relu=False, table=None, packed=None):
    """Convenience forward: y = relu?(scale*conv(x) + shift + res); `scale` is folded into the packed weights."""
    Nb, _, H, W = x.shape
    OH, OW = spec.out_hw(H, W)
    order = gemm_order(spec, False)
    table = conv_table(spec, H, W, False, x.device, order) if table is None else table
    packed = conv_pack(spec, weights, False, scale, order=order) if packed is None else packed
    out = torch.empty((Nb, spec.cout, OH, OW), dtype=torch.float32, device=x.device)
    return conv_gemm(x, packed, table, out, (OH, OW), spec.stride, spec.cout, spec.K, 1, shift, res, None, relu)


def conv_dgrad(spec, dz, weights, in_hw, scale=None, res=None, mask=None, table=None, packed=None):
    """dx = conv^T(dz * scale[co]) (+res) (masked).  stride 1 for any kernel; stride>1 only for 1x1."""
    Nb, _, OH, OW = dz.shape
    H, W = in_hw
    order = gemm_order(spec, True)
    table = conv_table(spec, OH, OW, True, dz.device, order) if table is None else table
    packed = conv_pack(spec, weights, True, scale, order=order) if packed is None else packed
    if spec.stride == 1:
        dx = torch.empty((Nb, spec.cin, H, W), dtype=torch.float32, device=dz.device)
        return conv_gemm(dz, packed, table, dx, (H, W), 1, spec.cin, spec.Kt, 1, None, res, mask, False)
    assert spec.taps == 1 and spec.branches[0][3] == 0, "strided data-gradient only for 1x1 convolutions"
    # scatter onto the stride lattice.  Positions off the lattice keep `res` (or 0): the accumulation
    # runs IN PLACE on `res` (each element is read and written by the same thread).
    if res is None:
        dx = torch.empty((Nb, spec.cin, H, W), dtype=torch.float32, device=dz.device)
        dx.zero_()
    else:
        dx = res
    assert not isinstance(mask, ReluBits), "strided data gradient: the pattern is applied by relu_mask on the fp32 activation"
    conv_gemm(dz, packed, table, dx, (OH, OW), 1, spec.cin, spec.Kt, spec.stride, None,
              dx if res is not None else None, None, False)
    return relu_mask(dx, mask) if mask is not None else dx


def dot_rows(spec):
    """Rows of the partial d-gamma dot term `conv_wgrad(dot=...)` fills (one per block of 64 input channels)."""
    return L.load().dasac_conv_wgrad_dot_rows(spec.cin, spec.taps)


def conv_wgrad(spec, dz, x, weights, scale=None, dot=None, table=None, sum_dz=None, outs=None):
    """Returns [dW per branch]; optionally fills dot [dot_rows(spec), Cout] with the partial rows of sum_k W*G (unscaled G;
    `bn_param_grads` adds them in a fixed order -- no atomics) and sum_dz[co] = sum over batch and pixels of dz.
    `outs`: destination per branch (None entries are allocated) -- a gradient sink hands out slices of its flat
    reduction buffer here."""
    lib = L.load()
    L.require_gpu(dz, x, *weights)
    assert dot is None or (tuple(dot.shape) == (dot_rows(spec), spec.cout) and dot.is_contiguous())
    Nb, Cx, H, W = x.shape
    _, M, OH, OW = dz.shape
    table = conv_table(spec, H, W, False, x.device) if table is None else table
    nbytes = lib.dasac_conv_wgrad_workspace(Nb, OH, OW, M, spec.K)
    ws = L.workspace(nbytes, x.device)
    with PROFILE.span("conv_wgrad", 2.0 * Nb * OH * OW * M * spec.K, (M, spec.K, Nb * OH * OW, spec.stride, 1, False, False),
                      4.0 * (dz.numel() + x.numel() + M * spec.K)):
        fn = lib.dasac_conv_wgrad_x3 if PRECISION == "bf16x3" else lib.dasac_conv_wgrad
        L.check(fn(_c(dz).data_ptr(), x.data_ptr(), table.data_ptr(), Nb, Cx, H, W, OH, OW, spec.stride, M,
                   spec.K, ws.data_ptr(), ws.numel(), L.stream_ptr()), "dasac_conv_wgrad")
    grads, tap0 = [], 0
    for bi, (w, (kh, kw, _, _)) in enumerate(zip(weights, spec.branches)):
        dw = _dest(None if outs is None else outs[bi], w)
        L.check(lib.dasac_conv_wgrad_finish(ws.data_ptr(), Nb, OH, OW, M, spec.K, _c(w).data_ptr(), L.ptr(scale),
                                            dw.data_ptr(), L.ptr(dot), L.ptr(sum_dz if tap0 == 0 else None), spec.cin, kh * kw,
                                            tap0, L.stream_ptr()),
                "dasac_conv_wgrad_finish")
        grads.append(dw)
        tap0 += kh * kw
    return grads


# ----------------------------------------------------------------------------------------------
# SAC head
# ----------------------------------------------------------------------------------------------
def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _dest(out, like):
    """`out` if given (checked: fp32, contiguous, same element count as `like`), else a fresh tensor shaped like `like`."""
    if out is None:
        return torch.empty_like(like, memory_format=torch.contiguous_format)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == like.numel() and out.device == like.device
    return out


def upsample_softmax(logits, size, ignore=None, want_up=True, want_probs=False, want_sums=False):
    """bilinear(ac=True) upsampling [+ masked softmax + class sums].  Returns (up, probs, class_sums)."""
    lib = L.load()
    L.require_gpu(logits, ignore)
    logits = _c(logits)
    B, Cn, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    up = _f32((B, Cn, H, W), logits) if want_up else None
    probs = _f32((B, Cn, H, W), logits) if want_probs else None
    sums = torch.empty(Cn, dtype=torch.float64, device=logits.device) if want_sums else None
    ign = None if ignore is None else _c(ignore).view(torch.uint8)
    with PROFILE.span("upsample_softmax", 0.0, None, logits.numel() * 4.0 + B * Cn * H * W * 4.0 * (int(want_up) + int(want_probs)) + (B * H * W if ignore is not None else 0)):
        L.check(lib.dasac_upsample_softmax(logits.data_ptr(), B, Cn, h, w, H, W, L.ptr(ign), L.ptr(up), L.ptr(probs),
                                           L.ptr(sums), L.stream_ptr()), "dasac_upsample_softmax")
    return up, probs, sums


def infer_labels(logits, size, lut=None, want_conf=False):
    """infer_val.py:160-163 + writer: uint8 label map [B,H,W] = lut[argmax softmax(bilinear_ac(logits))] (+ winning prob)."""
    lib = L.load()
    L.require_gpu(logits, lut)
    logits = _c(logits)
    B, Cn, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    labels = torch.empty((B, H, W), dtype=torch.uint8, device=logits.device)
    conf = _f32((B, H, W), logits) if want_conf else None
    if lut is not None:
        assert lut.dtype == torch.uint8 and lut.numel() >= Cn and lut.is_contiguous()
    L.check(lib.dasac_infer_labels(logits.data_ptr(), B, Cn, h, w, H, W, L.ptr(lut), labels.data_ptr(), L.ptr(conf),
                                   L.stream_ptr()), "dasac_infer_labels")
    return labels, conf


def upsample_bwd(grad_up, low_hw, gscale=None):
    lib = L.load()
    L.require_gpu(grad_up, gscale)
    grad_up = _c(grad_up)
    B, Cn, H, W = grad_up.shape
    h, w = low_hw
    out = _f32((B, Cn, h, w), grad_up)
    nbytes = lib.dasac_upsample_bwd_workspace(B * Cn, H, w)
    ws = L.workspace(nbytes, grad_up.device)
    L.check(lib.dasac_upsample_bwd(grad_up.data_ptr(), B * Cn, h, w, H, W, L.ptr(gscale), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), L.stream_ptr()), "dasac_upsample_bwd")
    return out


def ce_loss(logits_up, labels, class_weight=None, conf=None, want_grad=False, want_per_class=False, gscale=None):
    """Returns (loss[1], dlogits or None, per_class or None); conf given -> focal_ce_conf broadcast form."""
    lib = L.load()
    L.require_gpu(logits_up, labels, class_weight, conf)
    logits_up, labels = _c(logits_up), _c(labels)
    B, Cn, H, W = logits_up.shape
    HW = H * W
    loss = _f32((1,), logits_up)
    dl = torch.empty_like(logits_up) if want_grad else None
    pc = _f32((Cn,), logits_up) if want_per_class else None
    nbytes = lib.dasac_ce_loss_workspace(B, Cn, HW)
    ws = L.workspace(nbytes, logits_up.device)
    with PROFILE.span("ce_loss", 0.0, None, logits_up.numel() * 4.0 * (2 if want_grad else 1) + B * HW * (8 + (4 if conf is not None else 0))):
        L.check(lib.dasac_ce_loss(logits_up.data_ptr(), labels.data_ptr(), L.ptr(class_weight), L.ptr(None if conf is None else _c(conf)),
                                  B, Cn, HW, 0 if conf is None else 1, L.ptr(gscale), loss.data_ptr(), L.ptr(dl), L.ptr(pc), ws.data_ptr(),
                                  ws.numel(), L.stream_ptr()), "dasac_ce_loss")
    return loss, dl, pc


def ce_loss_bwd_low(logits_up, labels, low_hw, class_weight=None, conf=None, gscale=None):
    """d loss / d (low-resolution logits) of `ce_loss` composed with the bilinear upsampling, in one pass over logits_up."""
    lib = L.load()
    L.require_gpu(logits_up, labels, class_weight, conf, gscale)
    logits_up, labels = _c(logits_up), _c(labels)
    B, Cn, H, W = logits_up.shape
    h, w = int(low_hw[0]), int(low_hw[1])
    out = _f32((B, Cn, h, w), logits_up)
    ws = L.workspace(lib.dasac_ce_loss_bwd_low_workspace(B, Cn, H, w), logits_up.device)
    with PROFILE.span("ce_loss_bwd_low", 0.0, None, logits_up.numel() * 4.0 + B * H * W * (8 + (4 if conf is not None else 0)) + out.numel() * 4.0):
        L.check(lib.dasac_ce_loss_bwd_low(logits_up.data_ptr(), labels.data_ptr(), L.ptr(class_weight), L.ptr(None if conf is None else _c(conf)),
                                          B, Cn, H, W, h, w, 0 if conf is None else 1, L.ptr(gscale), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          L.stream_ptr()), "dasac_ce_loss_bwd_low")
    return out


def warp_affine(x, theta):
    lib = L.load()
    L.require_gpu(x, theta)
    x, theta = _c(x), _c(theta)
    B, Cn, H, W = x.shape
    out = torch.empty_like(x)
    L.check(lib.dasac_warp_affine(x.data_ptr(), theta.data_ptr(), B, Cn, H, W, out.data_ptr(), L.stream_ptr()), "dasac_warp_affine")
    return out


POOL_MODES = {"avg_pool": 0, "minentropy_pool": 1}


def warp_pool(probs, theta, theta_inv, T, mode="avg_pool", tolerance=0.1, want_aligned=True):
    """probs [N*T,C,H,W] -> (pooled [N,C,H,W], mask [N,1,H,W], aligned or None).  theta None: the views are already
    aligned and coverage-weighted, only the pooling runs (SAC._avg_pool / _minentropy_pool called on their own)."""
    lib = L.load()
    L.require_gpu(probs, theta, theta_inv)
    probs = _c(probs)
    if theta is None:
        theta_inv, want_aligned = None, False
    else:
        theta, theta_inv = _c(theta), _c(theta_inv)
    NT, Cn, H, W = probs.shape
    assert NT % T == 0
    N = NT // T
    pooled = _f32((N, Cn, H, W), probs)
    mask = _f32((N, 1, H, W), probs)
    aligned = torch.empty_like(probs) if want_aligned else None
    with PROFILE.span("warp_pool", 0.0, None, probs.numel() * 4.0 * (2 if want_aligned else 1) + pooled.numel() * 4.0 + mask.numel() * 4.0):
        L.check(lib.dasac_warp_pool(probs.data_ptr(), L.ptr(theta), L.ptr(theta_inv), N, T, Cn, H, W, POOL_MODES[mode],
                                    float(tolerance), L.ptr(aligned), pooled.data_ptr(), mask.data_ptr(), L.stream_ptr()),
                "dasac_warp_pool")
    return pooled, mask, aligned


def warp_back(pooled, mask, theta_inv, views_per_group):
    lib = L.load()
    L.require_gpu(pooled, mask, theta_inv)
    theta_inv = _c(theta_inv)
    N, Cn, H, W = pooled.shape
    B = theta_inv.shape[0]
    assert B == N * views_per_group
    out = _f32((B, Cn, H, W), pooled)
    with PROFILE.span("warp_back", 0.0, None, (pooled.numel() + mask.numel() + out.numel()) * 4.0):
        L.check(lib.dasac_warp_back(pooled.data_ptr(), mask.data_ptr(), theta_inv.data_ptr(), B, views_per_group, Cn, H, W,
                                    out.data_ptr(), L.stream_ptr()), "dasac_warp_back")
    return out


def class_state(running_conf, class_sums, B, HW, beta, stat_momentum, update, focal_p, want_disc=True, want_focal=True):
    """In-place update of running_conf (when update) and the derived (disc, focal) vectors."""
    lib = L.load()
    L.require_gpu(running_conf, class_sums)
    Cn = running_conf.numel()
    disc = _f32((Cn,), running_conf) if want_disc else None
    focal = _f32((Cn,), running_conf) if want_focal else None
    L.check(lib.dasac_class_state(running_conf.data_ptr(), L.ptr(class_sums), int(B), int(HW), Cn, float(beta),
                                  float(stat_momentum), int(bool(update)), float(focal_p), L.ptr(disc), L.ptr(focal),
                                  L.stream_ptr()), "dasac_class_state")
    return disc, focal


class HostClassVectors:
    """disc = 1 - exp(-chi/beta) (sac.py:151-152) and focal = (1 - clamp(chi, 0))**p (sac.py:120,135) computed by the
    SAME CPU ATen kernels the reference runs (true division by the python scalar, Sleef exp/pow) so that the per-class
    thresholds -- and with them the integer label map -- are bit-equal to the CPU reference's on equal probabilities;
    the device's expf/powf differ from Sleef's in the last bit on some inputs.

    Cost: 19 floats D2H + 2x19 H2D per target step.  `start()` queues the copy of chi into pinned memory right
    after the kernel that updates it; `finish()` (called after the warp/pool kernels were queued, so the GPU has
    work while the host waits) runs the two formulas on the host and uploads them without blocking."""

    _pinned = {}          # chi's storage -> (chi buffer, vectors buffer): pinned once per model, reused every step

    def __init__(self, running_conf):
        key = (running_conf.device.index, running_conf.data_ptr(), tuple(running_conf.shape))
        bufs = HostClassVectors._pinned.get(key)
        if bufs is None:
            bufs = (torch.empty(running_conf.shape, dtype=torch.float32).pin_memory(),
                    torch.empty((2,) + tuple(running_conf.shape), dtype=torch.float32).pin_memory())
            HostClassVectors._pinned[key] = bufs
        self.host, self.vecs = bufs
        self.host.copy_(running_conf.detach(), non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record()
        self.device = running_conf.device

    def finish(self, beta, focal_p, want_disc=True):
        self.done.synchronize()
        chi, vecs = self.host, self.vecs
        # (the previous step's upload from `vecs` finished long ago: every step waits on `done` after queueing it)
        vecs[0] = 1 - torch.exp(-chi / beta) if want_disc else 1.0
        vecs[1] = (1 - chi.clamp(0.)) ** focal_p
        dev = vecs.to(self.device, non_blocking=True)
        return (dev[0] if want_disc else None), dev[1]


class DeviceClassVectors:
    """The same two vectors from `dasac_class_state` on the device (exp through the fp64 library, rounded once): no copy to the
    host, no synchronisation -- a step that uses them is one uninterrupted launch sequence.  Within 1 ULP of HostClassVectors,
    not bit-equal to it by construction (see there), so pseudo labels can differ from the CPU reference's on a pixel whose
    confidence equals a threshold to the last bit.  Opt-in: `SAC.device_thresholds = True`."""

    def __init__(self, disc, focal):
        self.disc, self.focal = disc, focal

    def finish(self, beta, focal_p, want_disc=True):
        return (self.disc if want_disc else None), self.focal


def class_vectors(running_conf, beta, focal_p, want_disc=True):
    """One-shot form of HostClassVectors (blocks until the stream reaches this point)."""
    L.require_gpu(running_conf)
    return HostClassVectors(running_conf).finish(beta, focal_p, want_disc)


# ----------------------------------------------------------------------------------------------
# around the convolutions
# ----------------------------------------------------------------------------------------------
def bn_fold(gamma, beta, mean, var, eps, conv_bias=None, want_invstd=True):
    lib = L.load()
    L.require_gpu(gamma, beta, mean, var, conv_bias)
    Cn = gamma.numel()
    scale, shift = _f32((Cn,), gamma), _f32((Cn,), gamma)
    invstd = _f32((Cn,), gamma) if want_invstd else None
    L.check(lib.dasac_bn_fold(gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), L.ptr(conv_bias), float(eps),
                              Cn, scale.data_ptr(), shift.data_ptr(), L.ptr(invstd), L.stream_ptr()), "dasac_bn_fold")
    return scale, shift, invstd


def build_refresh_tables(fold_jobs, pack_jobs, device):
    """Device tables for dasac_bn_fold_multi / dasac_conv_pack_multi (include/dasac_hip.h: dasac_fold_job, dasac_pack_job).
    fold job: (gamma, beta, mean, var, conv_bias|None, (scale, shift, invstd), eps, C)
    pack job: (weight, scale|None, out, Cout, Cin, taps, Mpad, Kpad, mode, order)"""
    import numpy as np
    lib = L.load()
    fold_dt = np.dtype([("gamma", "<u8"), ("beta", "<u8"), ("mean", "<u8"), ("var", "<u8"), ("conv_bias", "<u8"), ("scale", "<u8"),
                        ("shift", "<u8"), ("invstd", "<u8"), ("eps", "<f4"), ("C", "<i4")])
    pack_dt = np.dtype([("w", "<u8"), ("scale", "<u8"), ("out", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("taps", "<i4"), ("Mpad", "<i4"),
                        ("Kpad", "<i4"), ("mode", "<i4"), ("order", "<i4"), ("reserved", "<i4")])
    assert fold_dt.itemsize == 72 and pack_dt.itemsize == 56
    keep = []
    fj = np.zeros(len(fold_jobs), dtype=fold_dt)
    fchunks = []
    for i, (g, b, m, v, cb, outs, eps, C) in enumerate(fold_jobs):
        L.require_gpu(g, b, m, v, cb, *outs)
        fj[i] = (g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), 0 if cb is None else cb.data_ptr(), outs[0].data_ptr(),
                 outs[1].data_ptr(), outs[2].data_ptr(), float(eps), int(C))
        fchunks += [(i, c) for c in range((C + 255) // 256)]
        keep.append(outs)
    pj = np.zeros(len(pack_jobs), dtype=pack_dt)
    pchunks, chunk = [], lib.dasac_pack_chunk_elems()
    for i, (w, sc, out, Cout, Cin, taps, Mpad, Kpad, mode, order) in enumerate(pack_jobs):
        L.require_gpu(w, sc, out)
        assert w.is_contiguous() and out.is_contiguous() and out.numel() == Kpad * Mpad
        pj[i] = (w.data_ptr(), 0 if sc is None else sc.data_ptr(), out.data_ptr(), Cout, Cin, taps, Mpad, Kpad, mode, order, 0)
        pchunks += [(i, c) for c in range((Kpad * Mpad + chunk - 1) // chunk)]
        keep.append(out)
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device) if a.size else None
    upi = lambda l: torch.tensor(l, dtype=torch.int32).reshape(-1, 2).to(device) if l else None
    return {"fold": up(fj), "fold_chunks": upi(fchunks), "n_fold_chunks": len(fchunks),
            "pack": up(pj), "pack_chunks": upi(pchunks), "n_pack_chunks": len(pchunks), "keep": keep}


def refresh_network(tables):
    """All frozen-BN folds, then all packed weight operands of a network: two launches."""
    lib = L.load()
    if tables["n_fold_chunks"]:
        L.check(lib.dasac_bn_fold_multi(tables["fold"].data_ptr(), tables["fold_chunks"].data_ptr(), tables["n_fold_chunks"],
                                        L.stream_ptr()), "dasac_bn_fold_multi")
    if tables["n_pack_chunks"]:
        L.check(lib.dasac_conv_pack_multi(tables["pack"].data_ptr(), tables["pack_chunks"].data_ptr(), tables["n_pack_chunks"],
                                          L.stream_ptr()), "dasac_conv_pack_multi")


def bn_param_grads(dot, sum_dz, mean, invstd, scale, conv_bias, want_gamma=True, want_beta=True, want_bias=False, outs=(None, None, None)):
    lib = L.load()
    Cn = sum_dz.numel()
    dg = _dest(outs[0], sum_dz) if want_gamma else None
    db = _dest(outs[1], sum_dz) if want_beta else None
    dcb = _dest(outs[2], sum_dz) if want_bias else None
    assert dot is None or (dot.dim() == 2 and dot.shape[1] == Cn and dot.is_contiguous())
    L.check(lib.dasac_bn_param_grads(L.ptr(dot), 0 if dot is None else dot.shape[0], sum_dz.data_ptr(), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(conv_bias),
                                     Cn, L.ptr(dg), L.ptr(db), L.ptr(dcb), L.stream_ptr()), "dasac_bn_param_grads")
    return dg, db, dcb


def channel_sums(x, out=None):
    lib = L.load()
    L.require_gpu(x)
    x = _c(x)
    N, Cn = x.shape[0], x.shape[1]
    out = _f32((Cn,), x) if out is None else out
    assert out.numel() == Cn and out.dtype == torch.float32 and out.is_contiguous()
    L.check(lib.dasac_channel_sums(x.data_ptr(), N, Cn, x[0, 0].numel(), out.data_ptr(), L.stream_ptr()), "dasac_channel_sums")
    return out


def pool_out(n, k, s, p, ceil_mode):
    """torch's pooling output-size rule."""
    num = n + 2 * p - k
    o = (-(-num // s) if ceil_mode else num // s) + 1
    if ceil_mode and (o - 1) * s >= n + p:
        o -= 1
    return o


def maxpool_fwd(x, k, s, p, ceil_mode):
    lib = L.load()
    L.require_gpu(x)
    x = _c(x)
    B, Cn, H, W = x.shape
    OH, OW = pool_out(H, k, s, p, ceil_mode), pool_out(W, k, s, p, ceil_mode)
    y = _f32((B, Cn, OH, OW), x)
    arg = torch.empty((B, Cn, OH, OW), dtype=torch.uint8, device=x.device)
    with PROFILE.span("maxpool_fwd", 0.0, None, x.numel() * 4.0 + y.numel() * 5.0):
        L.check(lib.dasac_maxpool_fwd(x.data_ptr(), B * Cn, H, W, OH, OW, k, s, p, y.data_ptr(), arg.data_ptr(), L.stream_ptr()),
                "dasac_maxpool_fwd")
    return y, arg


def maxpool_bwd(dy, y, arg, in_hw, k, s, p, relu_mask):
    lib = L.load()
    L.require_gpu(dy, y, arg)
    dy = _c(dy)
    B, Cn, OH, OW = dy.shape
    H, W = in_hw
    dx = _f32((B, Cn, H, W), dy)
    # algorithmic bytes: dx written, dy + the argmax byte read; the pooled tensor only for windows beyond 11x11 (bit 7 of the byte)
    with PROFILE.span("maxpool_bwd", 0.0, None, dx.numel() * 4.0 + dy.numel() * (9.0 if (relu_mask and k > 11) else 5.0)):
        L.check(lib.dasac_maxpool_bwd(dy.data_ptr(), y.data_ptr(), arg.data_ptr(), B * Cn, H, W, OH, OW, k, s, p, int(relu_mask),
                                      dx.data_ptr(), L.stream_ptr()), "dasac_maxpool_bwd")
    return dx


def add(a, b, out=None):
    lib = L.load()
    L.require_gpu(a, b)
    out = torch.empty_like(a) if out is None else out
    L.check(lib.dasac_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), L.stream_ptr()), "dasac_add")
    return out


def relu_mask(dy, y):
    lib = L.load()
    L.require_gpu(dy, y)
    out = torch.empty_like(dy)
    L.check(lib.dasac_relu_mask(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), L.stream_ptr()), "dasac_relu_mask")
    return out


def scale_planes(x, plane_scale):
    lib = L.load()
    L.require_gpu(x, plane_scale)
    x = _c(x)
    out = torch.empty_like(x)
    L.check(lib.dasac_scale_planes(x.data_ptr(), _c(plane_scale).data_ptr(), x.shape[0] * x.shape[1], x[0, 0].numel(),
                                   out.data_ptr(), L.stream_ptr()), "dasac_scale_planes")
    return out


def label_pad_mask(labels, pad_label=-1, ignore_label=255):
    """sac.py:337-338: returns ignore_mask = (labels == -1) (bool, same shape) and rewrites those labels to 255 IN PLACE."""
    lib = L.load()
    L.require_gpu(labels)
    assert labels.dtype == torch.int64
    work = labels if labels.is_contiguous() else labels.contiguous()
    mask = torch.empty(labels.shape, dtype=torch.bool, device=labels.device)
    L.check(lib.dasac_label_pad_mask(work.data_ptr(), mask.data_ptr(), work.numel(), int(pad_label), int(ignore_label),
                                     L.stream_ptr()), "dasac_label_pad_mask")
    if work is not labels:
        labels.copy_(work)                 # a strided caller tensor still sees the in-place rewrite
    else:
        torch.autograd.graph.increment_version(labels)
    return mask


def dropout_planes(B, Cn, p, device):
    """Dropout2d noise [B, C]: keep/(1-p), drawn on the device by a counter-based Philox keyed on torch's CUDA generator.
    (seed, offset) come from the device generator's own Philox state and the offset is ADVANCED there, so
    torch.manual_seed / torch.cuda.get_rng_state / set_rng_state (checkpoint resume) reproduce the masks exactly as they
    do for ATen's dropout, per device.  (The numbers are not ATen's -- parity tests inject `module.keep_mask`.)"""
    lib = L.load()
    out = torch.empty((B, Cn), dtype=torch.float32, device=device)
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = gen.initial_seed(), gen.get_offset()
    gen.set_offset(offset + 4 * ((B * Cn + 3) // 4))             # Philox offsets move in multiples of 4
    L.check(lib.dasac_dropout_planes(seed & 0xFFFFFFFFFFFFFFFF, offset, float(p), B * Cn, out.data_ptr(), L.stream_ptr()),
            "dasac_dropout_planes")
    return out


def class_sums(probs):
    """float64 [C] sums over batch and pixels of probs [B,C,H,W] (sac.py:108 before the division)."""
    lib = L.load()
    L.require_gpu(probs)
    probs = _c(probs)
    B, Cn = probs.shape[0], probs.shape[1]
    return _bn_sums(lib.dasac_bn_stats, (probs.data_ptr(),), B, Cn, probs[0, 0].numel(), probs.device, "dasac_bn_stats")[:Cn]


def _bn_sums(fn, lead_ptrs, N, Cn, HW, device, what):
    """Two-stage per-channel reduction (dasac_bn_stats / dasac_bn_bwd_reduce): float64 [2*C]."""
    lib = L.load()
    sums = torch.empty(2 * Cn, dtype=torch.float64, device=device)
    ws = L.workspace(lib.dasac_bn_stats_workspace(N, Cn, HW), device)
    L.check(fn(*lead_ptrs, N, Cn, HW, sums.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()), what)
    return sums


def bump_versions(tensors):
    """Kernels write parameters through raw pointers; autograd's version counters (which key the engine's packed-weight
    and BN-fold caches) have to be told."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


class EmaPlan:
    """Device-side pointer/chunk tables for the multi-tensor teacher update (built once; rebuilt by the
    caller when parameter storage moves)."""

    def __init__(self, fast_tensors, slow_tensors):
        import numpy as np
        lib = L.load()
        L.require_gpu(*fast_tensors)
        L.require_gpu(*slow_tensors)
        chunk = lib.dasac_ema_chunk_elems()
        pairs = np.zeros((len(fast_tensors), 3), dtype=np.int64)
        chunks = []
        for i, (f, s) in enumerate(zip(fast_tensors, slow_tensors)):
            assert f.is_contiguous() and s.is_contiguous() and f.numel() == s.numel() and f.dtype == torch.float32
            pairs[i] = (f.data_ptr(), s.data_ptr(), f.numel())
            chunks += [(i, j) for j in range((f.numel() + chunk - 1) // chunk)]
        dev = fast_tensors[0].device
        self.key = tuple(int(v) for v in pairs[:, :2].reshape(-1))
        self.pairs = torch.from_numpy(pairs).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self.sq = torch.empty(len(fast_tensors) + len(chunks), dtype=torch.float64, device=dev)   # per-tensor sums + chunk partials
        self.n_tensors, self.n_chunks = len(fast_tensors), len(chunks)
        self.slow = list(slow_tensors)

    def run(self, momentum, update):
        lib = L.load()
        out = torch.empty(1, dtype=torch.float32, device=self.pairs.device)
        L.check(lib.dasac_ema_update(self.pairs.data_ptr(), self.n_tensors, self.chunks.data_ptr(), self.n_chunks,
                                     float(momentum), int(bool(update)), self.sq.data_ptr(), out.data_ptr(), L.stream_ptr()),
                "dasac_ema_update")
        if update:
            bump_versions(self.slow)
        return out


# ----------------------------------------------------------------------------------------------
# train-mode BatchNorm (batch statistics; SyncBN when a process group with >1 ranks exists)
# ----------------------------------------------------------------------------------------------
def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _allreduce_sums(sums, count):
    """SyncBN: one all-reduce of the raw sums + the element count over RCCL (deeplabv2.py:15); identity on one rank.
    Returns (sums, host count or 0.0, device count or None) -- with several ranks the count stays on the device."""
    if _world() == 1:
        return sums, float(count), None
    import torch.distributed as dist
    packed = torch.cat([sums, torch.tensor([float(count)], dtype=torch.float64, device=sums.device)])
    dist.all_reduce(packed)
    return packed[:-1], 0.0, packed[-1:]


def bn_train_forward(z, bn, res=None, relu=False, update_running=True, tile_stats=None):
    """y = relu?(BN_batch(z) (+res)); returns (y, (mean, invstd, count, count_dev)).  Updates bn.running_* like ATen.
    tile_stats: the per-tile channel statistics the producing conv_gemm(stats=...) left -- then z is not read for them."""
    lib = L.load()
    L.require_gpu(z, res, tile_stats)
    N, Cn = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    scale, shift, mean, invstd = (_f32((Cn,), z) for _ in range(4))
    mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
    upd = update_running and bn.track_running_stats
    tail = (bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr() if upd else None, bn.running_var.data_ptr() if upd else None,
            bn.num_batches_tracked.data_ptr() if upd else None, float(mom), float(bn.eps), Cn, scale.data_ptr(), shift.data_ptr(),
            mean.data_ptr(), invstd.data_ptr(), L.stream_ptr())
    if tile_stats is not None and _world() == 1 and N * Cn < 65536:
        # one rank: ONE launch per BN layer -- every block adds its channel's tile statistics itself, then normalises its chunk
        count, count_dev = float(N * HW), None
        y = torch.empty_like(z)
        L.check(lib.dasac_bn_train_apply_tiles(z.data_ptr(), tile_stats.data_ptr(), tile_stats.shape[0], tile_stats.shape[2], count,
                                               tail[0], tail[1], tail[2], tail[3], tail[4], tail[5], tail[6], L.ptr(res), int(relu),
                                               N, Cn, HW, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), L.stream_ptr()),
                "dasac_bn_train_apply_tiles")
        if upd:
            bump_versions([bn.running_mean, bn.running_var, bn.num_batches_tracked])
        return y, (mean, invstd, count, count_dev)
    else:
        if tile_stats is not None:
            sums = torch.empty(2 * Cn, dtype=torch.float64, device=z.device)
            L.check(lib.dasac_bn_tile_stats_reduce(tile_stats.data_ptr(), tile_stats.shape[0], Cn, tile_stats.shape[2], sums.data_ptr(),
                                                   L.stream_ptr()), "dasac_bn_tile_stats_reduce")
        else:
            sums = _bn_sums(lib.dasac_bn_stats, (z.data_ptr(),), N, Cn, HW, z.device, "dasac_bn_stats")
        sums, count, count_dev = _allreduce_sums(sums, N * HW)
        L.check(lib.dasac_bn_train_finalize(sums.data_ptr(), count, L.ptr(count_dev), *tail), "dasac_bn_train_finalize")
    if upd:                                                    # written through raw pointers; eval-mode folds key on them
        bump_versions([bn.running_mean, bn.running_var, bn.num_batches_tracked])
    y = torch.empty_like(z)
    L.check(lib.dasac_bn_apply(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(res), int(relu), N, Cn, HW, y.data_ptr(),
                               L.stream_ptr()), "dasac_bn_apply")
    return y, (mean, invstd, count, count_dev)


def bn_train_backward(dy, z, stats, gamma, want_params=True, outs=(None, None)):
    """dy: gradient w.r.t. the BN output (ReLU mask already applied).  Returns (dz, dgamma, dbeta)."""
    lib = L.load()
    L.require_gpu(dy, z)
    mean, invstd, count, count_dev = stats
    N, Cn = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    dy = _c(dy)
    dg = _dest(outs[0], gamma) if want_params else None
    db = _dest(outs[1], gamma) if want_params else None
    ws = L.workspace(lib.dasac_bn_stats_workspace(N, Cn, HW), z.device)
    if _world() == 1 and count_dev is None and N * Cn < 65536:
        # one rank: reduction stage 1 + ONE kernel that adds the partials per block, forms dz and writes d gamma / d beta
        dz = torch.empty_like(z)
        L.check(lib.dasac_bn_bwd_fused(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), float(count),
                                       N, Cn, HW, dz.data_ptr(), L.ptr(dg), L.ptr(db), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                "dasac_bn_bwd_fused")
        return dz, dg, db
    # several ranks: the finish kernel also writes d gamma / d beta -- this rank's LOCAL sums (DDP averages them afterwards)
    sums = torch.empty(2 * Cn, dtype=torch.float64, device=z.device)
    L.check(lib.dasac_bn_bwd_reduce(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), N, Cn, HW, sums.data_ptr(),
                                    L.ptr(dg), L.ptr(db), ws.data_ptr(), ws.numel(), L.stream_ptr()), "dasac_bn_bwd_reduce")
    if _world() > 1:
        import torch.distributed as dist
        dist.all_reduce(sums)
    dz = torch.empty_like(z)
    L.check(lib.dasac_bn_bwd_apply(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                   sums.data_ptr(), float(count), L.ptr(count_dev), N, Cn, HW, dz.data_ptr(), None, None, L.stream_ptr()),
            "dasac_bn_bwd_apply")
    return dz, dg, db


# ----------------------------------------------------------------------------------------------
# tap-expanded convolution (ASPP classifiers): dense 1x1 GEMMs over taps*Cp channels + shift kernels
# ----------------------------------------------------------------------------------------------
class ExpandedConv:
    """Static description: `spec` (multi-branch, stride 1) -> 1x1 spec over E = taps*Cp expanded channels."""

    def __init__(self, spec):
        assert spec.stride == 1 and spec.taps <= 64
        self.spec = spec
        cp = spec.cout
        while (spec.taps * cp) % 16:
            cp += 1
        self.cp, self.E = cp, spec.taps * cp
        self.spec1 = ConvSpec(spec.cin, self.E, [(1, 1, 1, 0)], 1)
        self._cols = [torch.tensor(c, dtype=_i32) for c in zip(*spec.branches)]      # host arrays kh, kw, dil, pad

    def _branch_args(self):
        c = self._cols
        return c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), len(self.spec.branches)

    def pack(self, weights, transposed, out=None):
        lib = L.load()
        M = self.spec.cin if transposed else self.E
        K = self.E if transposed else self.spec.cin
        if out is None:
            out = torch.empty((lib.dasac_conv_kpad(K), lib.dasac_conv_mpad(M)), dtype=torch.float32, device=weights[0].device)
        tap0 = 0
        for w, (kh, kw, _, _) in zip(weights, self.spec.branches):
            L.check(lib.dasac_conv_pack_expanded(_c(w).data_ptr(), self.spec.cout, self.spec.cin, kh * kw, tap0, self.spec.taps,
                                                 self.cp, int(transposed), out.data_ptr(), L.stream_ptr()), "dasac_conv_pack_expanded")
            tap0 += kh * kw
        return _finish_pack(out, M, K)

    def forward(self, x, packed, table, bias):
        """x [B,Cin,H,W] -> out [B,Cout,H,W]."""
        lib = L.load()
        B, _, H, W = x.shape
        y = torch.empty((B, self.E, H, W), dtype=torch.float32, device=x.device)
        conv_gemm(x, packed, table, y, (H, W), 1, self.E, self.spec.cin)
        out = torch.empty((B, self.spec.cout, H, W), dtype=torch.float32, device=x.device)
        L.check(lib.dasac_tap_gather(y.data_ptr(), *self._branch_args(), self.cp, self.spec.cout, L.ptr(bias), B, H, W,
                                     out.data_ptr(), L.stream_ptr()), "dasac_tap_gather")
        return out

    def scatter(self, dout):
        lib = L.load()
        B, _, H, W = dout.shape
        d = torch.empty((B, self.E, H, W), dtype=torch.float32, device=dout.device)
        L.check(lib.dasac_tap_scatter(_c(dout).data_ptr(), *self._branch_args(), self.cp, self.spec.cout, B, H, W, d.data_ptr(),
                                      L.stream_ptr()), "dasac_tap_scatter")
        return d

    def wgrad(self, d, x, weights, table, outs=None):
        """Weight gradients of every branch from the expanded gradient d [B,E,H,W]."""
        lib = L.load()
        B, Cx, H, W = x.shape
        nbytes = lib.dasac_conv_wgrad_workspace(B, H, W, self.E, self.spec.cin)
        ws = L.workspace(nbytes, x.device)
        with PROFILE.span("conv_wgrad", 2.0 * B * H * W * self.E * self.spec.cin, (self.E, self.spec.cin, B * H * W, 1, 1, False, False)):
            fn = lib.dasac_conv_wgrad_x3 if PRECISION == "bf16x3" else lib.dasac_conv_wgrad
            L.check(fn(d.data_ptr(), x.data_ptr(), table.data_ptr(), B, Cx, H, W, H, W, 1, self.E, self.spec.cin,
                       ws.data_ptr(), ws.numel(), L.stream_ptr()), "dasac_conv_wgrad")
        grads, tap0 = [], 0
        for bi, (w, (kh, kw, _, _)) in enumerate(zip(weights, self.spec.branches)):
            dw = _dest(None if outs is None else outs[bi], w)
            L.check(lib.dasac_conv_wgrad_finish_expanded(ws.data_ptr(), B, H, W, self.E, self.spec.cin, dw.data_ptr(),
                                                         self.spec.cout, kh * kw, tap0, self.cp, L.stream_ptr()),
                    "dasac_conv_wgrad_finish_expanded")
            grads.append(dw)
            tap0 += kh * kw
        return grads

    def dgrad(self, d, packed_t, table_t, in_hw, res=None, mask=None):
        B = d.shape[0]
        H, W = in_hw
        dx = torch.empty((B, self.spec.cin, H, W), dtype=torch.float32, device=d.device)
        return conv_gemm(d, packed_t, table_t, dx, (H, W), 1, self.spec.cin, self.E, 1, None, res, mask, False)


def iou_counts(logits_up, gt, counts=None, ignore_index=255):
    """Accumulates per-class (tp, fp, fn) pixel counts of argmax(logits_up) vs gt into `counts` (int64 [3,C])."""
    lib = L.load()
    L.require_gpu(logits_up, gt)
    logits_up, gt = _c(logits_up), _c(gt)
    B, Cn, H, W = logits_up.shape
    if counts is None:
        counts = torch.zeros((3, Cn), dtype=torch.int64, device=logits_up.device)
    L.check(lib.dasac_iou_counts(logits_up.data_ptr(), gt.data_ptr(), B, Cn, H * W, int(ignore_index), counts.data_ptr(),
                                 L.stream_ptr()), "dasac_iou_counts")
    return counts
