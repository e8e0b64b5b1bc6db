"""Test-only helpers: move tensors between the reference's NCHW layout (CPU, torch fp32 = the
oracle side) and the library's padded NHWC layout (GPU side)."""
import torch


def ceil4(c):
    return (c + 3) // 4 * 4


def to_dev_nhwc(x_nchw, dev):
    B, C, H, W = x_nchw.shape
    out = torch.zeros(B, H, W, ceil4(C), dtype=torch.float32)
    out[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return out.to(dev)


def from_dev_nhwc(y, C):
    return y[..., :C].permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol=1e-4, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max-abs error / max-abs reference = {e:.3e} > {tol}"
