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
