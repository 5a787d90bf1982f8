"""Tensor-level wrappers over the C ABI (one function per kernel family)."""
import torch

from . import lib as L


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


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


def conv_table(spec, plane_h, plane_w, transposed, device):
    """Gather table for planes of plane_h x plane_w (forward: the input planes; transposed: dz planes)."""
    lib = L.load()
    C = spec.cout if transposed else spec.cin
    K = spec.taps * C
    table = torch.empty((lib.dasac_conv_kpad(K), 4), dtype=_i32, device=device)
    cols = [torch.tensor(c, dtype=_i32) for c in zip(*spec.branches)]   # host arrays: kh, kw, dil, pad
    L.check(lib.dasac_conv_table(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), cols[3].data_ptr(),
                                 len(spec.branches), C, plane_h, plane_w, int(transposed), table.data_ptr(),
                                 L.stream_ptr()), "dasac_conv_table")
    return table


def conv_pack(spec, weights, transposed, scale=None, out=None):
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
                                    int(transposed), out.data_ptr(), L.stream_ptr()), "dasac_conv_pack")
        tap0 += kh * kw
    return out


def conv_gemm(x, packed, table, out, grid_hw, stride, M, K, ostride=1, scale=None, shift=None, res=None, mask=None,
              relu=False):
    """out[n,m,oh*os,ow*os] = epilogue(sum_k packed[k][m] * gather(x)); see dasac_conv_gemm."""
    lib = L.load()
    L.require_gpu(x, packed, table, out)
    Nb, Cx, H, W = x.shape
    OH, OW = grid_hw
    assert out.shape[0] == Nb and out.shape[1] == M and x.is_contiguous() and out.is_contiguous()
    for t_ in (res, mask):
        assert t_ is None or (t_.shape == out.shape and t_.is_contiguous())
    L.check(lib.dasac_conv_gemm(x.data_ptr(), packed.data_ptr(), table.data_ptr(), out.data_ptr(), Nb, Cx, H, W, OH, OW,
                                stride, M, K, out.shape[2], out.shape[3], ostride, L.ptr(scale), L.ptr(shift), L.ptr(res),
                                L.ptr(mask), int(relu), L.stream_ptr()), "dasac_conv_gemm")
    return out


def conv_forward(spec, x, weights, scale=None, shift=None, res=None, relu=False, table=None, packed=None):
    """Convenience forward: y = epi(conv(x))."""
    Nb, _, H, W = x.shape
    OH, OW = spec.out_hw(H, W)
    table = conv_table(spec, H, W, False, x.device) if table is None else table
    packed = conv_pack(spec, weights, False) if packed is None else packed
    out = torch.empty((Nb, spec.cout, OH, OW), dtype=torch.float32, device=x.device)
    return conv_gemm(x, packed, table, out, (OH, OW), spec.stride, spec.cout, spec.K, 1, scale, shift, res, None, relu)


def conv_dgrad(spec, dz, weights, in_hw, scale=None, res=None, mask=None, table=None, packed=None):
    """dx = conv^T(dz * scale[co]) (+res) (masked).  stride 1 for any kernel; stride>1 only for 1x1."""
    Nb, _, OH, OW = dz.shape
    H, W = in_hw
    table = conv_table(spec, OH, OW, True, dz.device) if table is None else table
    packed = conv_pack(spec, weights, True, scale) if packed is None else packed
    if spec.stride == 1:
        dx = torch.empty((Nb, spec.cin, H, W), dtype=torch.float32, device=dz.device)
        return conv_gemm(dz, packed, table, dx, (H, W), 1, spec.cin, spec.Kt, 1, None, None, res, mask, False)
    assert spec.taps == 1 and spec.branches[0][3] == 0, "strided data-gradient only for 1x1 convolutions"
    dx = torch.empty((Nb, spec.cin, H, W), dtype=torch.float32, device=dz.device)
    if res is None:
        dx.zero_()
    else:
        dx.copy_(res)
    # scatter onto the stride lattice; positions off the lattice keep res (or 0)
    return conv_gemm(dz, packed, table, dx, (OH, OW), 1, spec.cin, spec.Kt, spec.stride, None, None,
                     dx if res is not None else None, mask, False)


def conv_wgrad(spec, dz, x, weights, scale=None, dot=None, table=None):
    """Returns [dW per branch]; optionally accumulates dot[co] += sum_k W*G (unscaled G)."""
    lib = L.load()
    L.require_gpu(dz, x, *weights)
    Nb, Cx, H, W = x.shape
    _, M, OH, OW = dz.shape
    table = conv_table(spec, H, W, False, x.device) if table is None else table
    nbytes = lib.dasac_conv_wgrad_workspace(Nb, OH, OW, M, spec.K)
    ws = L.workspace(nbytes, x.device)
    L.check(lib.dasac_conv_wgrad(_c(dz).data_ptr(), x.data_ptr(), table.data_ptr(), Nb, Cx, H, W, OH, OW, spec.stride, M,
                                 spec.K, ws.data_ptr(), ws.numel(), L.stream_ptr()), "dasac_conv_wgrad")
    grads, tap0 = [], 0
    for w, (kh, kw, _, _) in zip(weights, spec.branches):
        dw = torch.empty_like(w)
        L.check(lib.dasac_conv_wgrad_finish(ws.data_ptr(), Nb, OH, OW, M, spec.K, _c(w).data_ptr(), L.ptr(scale),
                                            dw.data_ptr(), L.ptr(dot), spec.cin, kh * kw, tap0, L.stream_ptr()),
                "dasac_conv_wgrad_finish")
        grads.append(dw)
        tap0 += kh * kw
    return grads
