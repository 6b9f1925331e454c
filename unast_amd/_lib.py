"""ctypes loader for libunast_hip.so.  include/unast_hip.h is the single source of truth: argument types are
parsed from its prototypes, so every declared symbol must be exported by the library (checked at load)."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNAST_HIP_LIB") or os.path.join(_HERE, "libunast_hip.so")   # override: kernel experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "unast_hip.h")

_lib = None
_protos = None


class UnastHipError(RuntimeError):
    pass


def _ctype(decl):
    d = decl.strip()
    if "*" in d or "hipStream_t" in d:
        return ctypes.c_void_p
    base = re.sub(r"\b\w+$", "", d).strip() if len(d.split()) > 1 else d
    base = base.replace("const", "").strip()
    table = {"int": ctypes.c_int, "unsigned int": ctypes.c_uint, "uint32_t": ctypes.c_uint, "float": ctypes.c_float,
             "double": ctypes.c_double, "int64_t": ctypes.c_longlong, "long long": ctypes.c_longlong,
             "uint64_t": ctypes.c_ulonglong, "size_t": ctypes.c_size_t}
    if base not in table:
        raise ValueError("unast_hip.h: cannot map parameter %r" % decl)
    return table[base]


def parse_header(path=HEADER_PATH):
    """Returns {name: (restype, [argtypes])} for every prototype in the header."""
    global _protos
    if _protos is not None:
        return _protos
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int|void)\s+(unast_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (None if ret == "void" else (ctypes.c_longlong if ret == "int64_t" else ctypes.c_int))
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    _protos = protos
    return protos


def lib():
    """Loads the HIP library; fails loudly when it is missing (there is no CPU fallback)."""
    return ctypes_lib()


def ctypes_lib():
    """ctypes binding of libunast_hip.so with argument types parsed from the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnastHipError("libunast_hip.so not found at %s — build it with `make -C unast_amd/csrc` "
                            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path." % LIB_PATH)
    # The library's kernels register with the HIP runtime when it is loaded.  Loading it before the process's runtime (torch's) has
    # opened the device left them unlaunchable ("no ROCm-capable device is detected": build() followed by smoke() in one process), so the
    # device is opened first wherever there is one.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise UnastHipError("libunast_hip.so does not export %s declared in include/unast_hip.h" % name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


CAPTURE_NOTE = None        # while a HIP-graph capture is open: a callable that tells the library which stream the last launch went to (engine.capture_begins)


def check(status, what=""):
    if status != 0:
        raise UnastHipError("%s failed (%d): %s" % (what or "unast call", status, lib().unast_last_error().decode()))
    if CAPTURE_NOTE is not None:
        CAPTURE_NOTE()
