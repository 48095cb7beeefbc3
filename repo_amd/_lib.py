"""ctypes binding of librepo_hip.so, generated from include/repo_hip.h.

The product path has no CPU fallback: if the library is missing or fails to load, every
op raises.  ``lib()`` loads lazily so that host-only logic (replay buffer, config,
data-parallel plumbing) imports without a GPU.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "repo_hip.h")
# REPO_HIP_LIB: developer knob -- load another build of the library (A/B runs of kernel variants in one process tree)
LIB_PATH = os.environ.get("REPO_HIP_LIB") or os.path.join(HERE, "librepo_hip.so")

_CT = {
    "int": ctypes.c_int,
    "unsigned": ctypes.c_uint,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "hipStream_t": ctypes.c_void_p,
}


class RepoHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every prototype."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|uint64_t|const char\s*\*)\s+(repo_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.+?)(\w+)$", a)
                params.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret.replace(" ", ""), params)
    # every declaration must have been understood: a prototype the pattern above fails to match would silently
    # lose its argtypes (ctypes would then pass 64-bit pointers as C ints)
    declared = set(re.findall(r"\b(repo_\w+)\s*\(", txt))
    if declared != set(protos):
        raise RepoHipError(f"include/repo_hip.h: unparsed prototypes {sorted(declared ^ set(protos))}")
    return protos


def _ctype(t):
    t = t.strip()
    if "*" in t:
        return ctypes.c_void_p
    return _CT[t.replace("const ", "").strip()]


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RepoHipError(
            f"{LIB_PATH} not found: build it with `python -m repo_amd.build` "
            "(the HIP path has no CPU fallback)"
        )
    # PyTorch-ROCm ships its own HIP runtime; it must be in the process BEFORE this library is
    # dlopen'ed so that both share one runtime (device pointers and streams come from torch).
    # Loading librepo_hip.so first binds it to the system runtime and every launch then fails
    # with hipErrorNoDevice.
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, params) in _protos.items():
        fn = getattr(L, name)
        fn.argtypes = [_ctype(t) for t, _ in params]
        fn.restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64,
                      "constchar*": ctypes.c_char_p}[ret]
    if L.repo_abi_version() != 8:
        raise RepoHipError("librepo_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().repo_strerror(rc).decode()
        raise RepoHipError(f"{what} failed: {msg} (code {rc})")
