"""Spatial transformer op -- mirror of /root/reference/air/transformer.py:18-175.

``transformer(U, theta, out_size)`` keeps the reference signature; U is
[B, H, W, 1] (or [B, H, W]) float32 on the GPU, theta [B, 6] or [B, 2, 3].
Runs the hand-written HIP kernel (air_transformer_fwd); no CPU fallback."""
import ctypes as C

import torch

from . import _hip as H


def transformer(U, theta, out_size, name="SpatialTransformer", **kwargs):
    if not U.is_cuda:
        raise H.AirHipError("transformer: U must be a device tensor (no CPU fallback)")
    squeeze = U.dim() == 4
    if squeeze and U.shape[3] != 1:
        raise NotImplementedError("only num_channels == 1 is on the AIR path (air_model.py:331, 364)")
    B, Hi, Wi = int(U.shape[0]), int(U.shape[1]), int(U.shape[2])
    Ho, Wo = int(out_size[0]), int(out_size[1])
    Uc = U.reshape(B, Hi, Wi).contiguous().float()
    th = theta.reshape(B, 6).contiguous().float()
    out = torch.empty(B, Ho, Wo, dtype=torch.float32, device=U.device)
    s = C.c_void_p(torch.cuda.current_stream(U.device).cuda_stream)
    H.check(H.lib().air_transformer_fwd(C.c_void_p(Uc.data_ptr()), C.c_void_p(th.data_ptr()),
                                        C.c_void_p(out.data_ptr()), B, Hi, Wi, Ho, Wo, s), "air_transformer_fwd")
    return out.unsqueeze(3) if squeeze else out
