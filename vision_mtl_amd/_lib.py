"""ctypes binding of libvmtl.so (the HIP kernels + C ABI declared in include/vmtl.h).

The header is the single source of truth: prototypes are parsed from it at import
time, so a function declared there but missing from the .so fails loudly here (and in
tests/test_abi.py).  There is NO fallback path: if the library cannot be loaded the
product raises — the hot path never silently runs anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from pathlib import Path

# torch first: PyTorch-ROCm ships its own libamdhip64.so, and libvmtl.so must bind to THAT copy of the HIP
# runtime (streams and device pointers come from torch).  Loaded the other way round, the dynamic loader
# resolves libvmtl.so against /opt/rocm's runtime and its launches fail with "no ROCm-capable device".
import torch  # noqa: F401

_PKG = Path(__file__).resolve().parent
CSRC = _PKG / "csrc"
LIB_PATH = Path(os.environ["VMTL_LIB"]) if os.environ.get("VMTL_LIB") else CSRC / "libvmtl.so"  # VMTL_LIB: A/B another build
HEADER = _PKG.parent / "include" / "vmtl.h"

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
}


def parse_header(path: Path = HEADER) -> dict:
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in vmtl.h."""
    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"(const char\*|long long|int)\s+(vmtl_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "const char*": ctypes.c_char_p}[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    ty, nm = a.rsplit(" ", 1)
                    argtypes.append(_CTYPES[ty.replace("const ", "").strip()])
                    argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


def build(force: bool = False) -> Path:
    """Compile libvmtl.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  Serialised with a file
    lock: the ranks of one node (one process per GPU) may all find the library missing at the same time."""
    import fcntl

    with open(CSRC / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.run(["make", "-C", str(CSRC), "clean"], check=True, capture_output=True)
            r = subprocess.run(["make", "-C", str(CSRC), "-j8"], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"building libvmtl.so failed:\n{r.stdout}\n{r.stderr}")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class _Lib:
    def __init__(self):
        if not LIB_PATH.exists():
            if os.environ.get("VMTL_NO_AUTOBUILD"):
                raise RuntimeError(f"{LIB_PATH} is missing; run `python -c 'import __graft_entry__ as g; g.build()'`")
            build()
        try:
            self._dll = ctypes.CDLL(str(LIB_PATH))
        except OSError as e:  # no fallback on purpose
            raise RuntimeError(f"cannot load the HIP extension {LIB_PATH}: {e}") from e
        self.protos = parse_header()
        self._fn = {}
        self._sig = {}  # (entry point, keyword tuple) -> (function, argument order): callk's validated call shapes
        for name, (restype, argtypes, _) in self.protos.items():
            try:
                f = getattr(self._dll, name)
            except AttributeError as e:
                raise RuntimeError(f"{LIB_PATH} does not export {name} declared in {HEADER}") from e
            f.restype = restype
            f.argtypes = argtypes
            self._fn[name] = f

    def raw(self, name):
        return self._fn[name]

    def call(self, name, *args):
        """Call an int-returning entry point; non-zero status becomes a Python exception."""
        rc = self._fn[name](*args)
        if rc != 0:
            self._raise(name, rc)

    def callk(self, name, **kw):
        """Keyword form: arguments are matched against the parameter NAMES of the prototype in
        vmtl.h (so a reordered or renamed C parameter is an immediate error, not silent
        corruption).  torch tensors become device pointers, None becomes NULL.
        The name check is done once per (entry point, keyword tuple) and cached: this is the eager launch path (~560
        calls per training step), where two set constructions per call were ~15 % of the host time."""
        sig = (name, tuple(kw))
        order = self._sig.get(sig)
        if order is None:
            _, _, argnames = self.protos[name]
            if set(kw) != set(argnames):
                missing, extra = set(argnames) - set(kw), set(kw) - set(argnames)
                raise TypeError(f"{name}: missing {sorted(missing)}, unexpected {sorted(extra)}")
            order = self._sig[sig] = (self._fn[name], tuple(argnames))
        fn, argnames = order
        args = []
        for n in argnames:
            v = kw[n]
            if v is not None and not isinstance(v, (int, float)):
                v = v.data_ptr()
            args.append(v)
        rc = fn(*args)
        if rc != 0:
            self._raise(name, rc)

    def _raise(self, name, rc):
        what = {-1: "bad argument", -2: "kernel launch failure", -3: "unsupported configuration"}.get(rc, "error")
        if rc == -2:
            what += ": " + self._fn["vmtl_last_error_string"]().decode()
        raise RuntimeError(f"{name} failed: {what} (status {rc})")


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
