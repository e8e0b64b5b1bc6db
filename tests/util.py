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


def assert_close(a, b, tol=1e-4, what="", atol=0.0):
    """max|a-b| <= tol * max|b| + atol.  atol is for quantities that are analytically zero (e.g. the
    gradient of a conv bias that feeds a train-mode BatchNorm), where both sides are rounding noise."""
    a, b = a.double(), b.double()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * ref + atol, f"{what}: max-abs error {err:.3e} vs reference magnitude {ref:.3e} (tol {tol}, atol {atol})"
