"""ctypes binding of libvvae_hip.so (the C ABI declared in include/vvae_hip.h).

The prototypes are parsed from the header so the binding cannot drift from the
declared ABI.  There is no CPU fallback: if the library is missing, or a
tensor is not on a GPU, the ops raise.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvvae_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vvae_hip.h")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
    "double": ctypes.c_double, "void": None,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|void)\s+(vvae_\w+)\s*\(([^)]*)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in args.split(","):
            a = a.strip()
            if not a or a == "void":
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                base = a.replace("const", "").split()[0]
                argtypes.append(_CT[base])
        protos[name] = (_CT[ret], argtypes)
    return protos


class VvaeError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (loads on first use).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VvaeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C video_vae_amd/csrc)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (ret, argtypes) in parse_header().items():
            fn = getattr(l, name)
            fn.restype = ret
            fn.argtypes = argtypes
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        raise VvaeError(f"{what} failed with status {code}"
                        + (" (bad argument)" if code == 1001 else " (workspace too small)" if code == 1002 else " (hipError_t)"))
