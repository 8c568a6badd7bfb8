"""bfloat16 <-> float32 helpers on raw uint16 bit patterns (NumPy has no bf16).

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np


def bf16_bits_to_f32(u16: np.ndarray) -> np.ndarray:
    """Exact widening: bf16 bit pattern -> float32 (AscendC Cast CAST_NONE bf16->f32)."""
    u = np.ascontiguousarray(u16).view(np.uint16).astype(np.uint32) << np.uint32(16)
    return u.view(np.float32)


def f32_to_bf16_bits_rne(f32: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit pattern, round-to-nearest-even (AscendC Cast CAST_RINT f32->bf16).

    NaN is mapped to the canonical quiet NaN 0x7FC0 (torch does the same).
    """
    x = np.ascontiguousarray(f32, dtype=np.float32).view(np.uint32)
    lsb = (x >> np.uint32(16)) & np.uint32(1)
    rounded = (x + np.uint32(0x7FFF) + lsb) >> np.uint32(16)
    out = rounded.astype(np.uint16)
    nan = np.isnan(np.ascontiguousarray(f32, dtype=np.float32))
    if nan.any():
        out = np.where(nan, np.uint16(0x7FC0), out)
    return out


def torch_to_bits(t):
    """torch bf16 tensor (cpu) -> uint16 ndarray of bit patterns."""
    import torch

    assert t.dtype == torch.bfloat16
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_torch(u16):
    import torch

    return torch.from_numpy(np.ascontiguousarray(u16).view(np.int16)).view(torch.bfloat16)
