"""-m "not gpu": the C-ABI library builds/loads on a GPU-less box and exports every symbol that
include/vmtl.h declares; the product path refuses CPU tensors instead of falling back."""
import ctypes

import pytest
import torch


def test_header_symbols_exported():
    from vision_mtl_amd._lib import HEADER, LIB_PATH, lib, parse_header

    protos = parse_header(HEADER)
    assert len(protos) >= 40
    dll = ctypes.CDLL(str(LIB_PATH)) if LIB_PATH.exists() else lib()._dll
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in vmtl.h but not exported"
    assert lib().raw("vmtl_version")().startswith(b"vmtl")


def test_host_only_entry_points():
    from vision_mtl_amd._lib import lib

    l = lib()
    assert l.raw("vmtl_reduce_rows")(1) == 1
    assert l.raw("vmtl_reduce_rows")(10 ** 6) == 1024
    assert l.raw("vmtl_conv2d_stats_rows")(32, 128, 256, 36) == 32 * 128 * 256 // l.raw("vmtl_conv2d_stats_block")(32, 128, 256, 36)
    assert l.raw("vmtl_conv2d_wgrad_splits")(32 * 128 * 256, 33, 612) >= 1
    assert l.raw("vmtl_ce_workspace_bytes")(1 << 20) % 8 == 0
    assert l.raw("vmtl_silog_workspace_bytes")(1 << 20) % 8 == 0


def test_no_cpu_fallback():
    from vision_mtl_amd import ops

    x = torch.zeros(1, 4, 4, 4)
    w = torch.zeros(4, 4, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(x, w, None, 1, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cross_entropy(torch.zeros(1, 3, 2, 2), torch.zeros(1, 2, 2, dtype=torch.int64))


def test_callk_rejects_wrong_names():
    from vision_mtl_amd._lib import lib

    with pytest.raises(TypeError):
        lib().callk("vmtl_maxpool2_fwd", x=None, y=None, B=1, H=2, W=2, Cs=4)  # stream missing


def test_torch_hip_runtime_is_loaded_before_the_extension():
    """libvmtl.so must resolve against the libamdhip64.so PyTorch-ROCm ships (streams / pointers come from
    torch): _lib imports torch before dlopen-ing it.  Loaded first, the extension binds to /opt/rocm's
    runtime and every launch fails with "no ROCm-capable device" (seen with build() followed by smoke()
    in one process).  Checked in a fresh interpreter via the order of the mappings in /proc/self/maps."""
    import subprocess
    import sys

    code = (
        "import re\n"
        "from vision_mtl_amd import _lib\n"
        "_lib.lib()\n"
        "maps = open('/proc/self/maps').read()\n"
        "hip = sorted(set(re.findall(r'(/\\S*libamdhip64[^\\s]*)', maps)))\n"
        "assert any('/torch/' in p for p in hip), hip\n"
        "assert not any(p.startswith('/opt/rocm') for p in hip), hip\n"
        "print('ok', hip)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    assert r.returncode == 0, r.stdout + r.stderr
