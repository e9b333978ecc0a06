"""Device-memory plumbing: torch owns HBM buffers and streams, libmpc_amd gets raw pointers."""
from __future__ import annotations

import numpy as np
import torch


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mpc_amd needs a ROCm GPU (gfx950); there is no CPU fallback for the control path")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device)
    return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())


def stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def f64(a, device: torch.device) -> torch.Tensor:
    """Contiguous float64 device tensor from a numpy array / sequence / tensor."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=device)


def i32(a, device: torch.device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.int32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=device)


def ptr(t) -> int | None:
    return None if t is None else int(t.data_ptr())
