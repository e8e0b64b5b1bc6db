"""-m gpu: the hand-written ctypes binding shown in INTEGRATION.md section 2 is executed as written (the code block is
read out of the document) and compared with torch: the documented ABI usage cannot rot."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_binding_runs_and_matches_torch(dev):
    import vision_mtl_amd._lib  # noqa: F401  (torch's HIP runtime first, then the library: see _lib.py)

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def conv_bn_relu" in b]
    assert len(blocks) == 1
    code = blocks[0].replace('"vision_mtl_amd/csrc/libvmtl.so"', repr(os.path.join(ROOT, "vision_mtl_amd", "csrc", "libvmtl.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 2, 21, 12, 20, 30
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(Cout)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
    x = torch.randn(B, Cin, H, W, generator=g)
    ref = F.relu(bn.train()(conv(x))).detach()
    rm_ref, rv_ref = bn.running_mean.clone(), bn.running_var.clone()
    bn2 = torch.nn.BatchNorm2d(Cout)
    bn2.load_state_dict({k: (v if "running" not in k and "num_batches" not in k else torch.nn.BatchNorm2d(Cout).state_dict()[k])
                         for k, v in bn.state_dict().items()})
    conv_d, bn_d = conv.to(dev), bn2.to(dev)
    Cs = (Cin + 3) // 4 * 4
    xd = torch.zeros(B, H, W, Cs, device=dev)
    xd[..., :Cin] = x.permute(0, 2, 3, 1).to(dev)
    out = ns["conv_bn_relu"](xd, conv_d, bn_d)
    torch.cuda.synchronize()
    got = out[..., :Cout].permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert float(out[..., Cout:].abs().max()) == 0.0 if out.shape[-1] > Cout else True
    assert torch.allclose(bn_d.running_mean.cpu(), rm_ref, atol=1e-5) and torch.allclose(bn_d.running_var.cpu(), rv_ref, atol=1e-5)
