"""ctypes binding of the C ABI declared in include/manta_hip.h.

The product library is ``mantaflow_amd/csrc/libmanta_hip.so`` (HIP, gfx950).  There is no CPU fallback: if the
library is missing or cannot be loaded the import of any hot-path function raises.  ``use_library`` exists so that
the test-suite can drive the *same* host code with another implementation of the same ABI (the plain-C oracle
restatement, host pointers) -- nothing in this package ever loads anything from ``oracle/`` by itself.

Prototypes are parsed from the header itself, so the binding cannot drift from ``include/manta_hip.h``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "manta_hip.h")
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libmanta_hip.so")

_CTYPES = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(mf_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes, argnames = [], []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                pm = re.match(r"(.*?)(\w+)$", a)
                typ, an = pm.group(1).strip(), pm.group(2)
                if "*" in typ:
                    argtypes.append(ctypes.c_char_p if "char" in typ else ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[typ.replace("const", "").strip()])
                argnames.append(an)
        protos[name] = (restype, argtypes, argnames)
    return protos


class Library:
    """A loaded implementation of the ABI; ``device`` is the torch device its pointers must live on."""

    def __init__(self, path, device):
        if not os.path.exists(path):
            raise RuntimeError(
                "mantaflow_amd: native library %s not found -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % path)
        self.path = path
        self.device = device
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        missing = []
        for name, (restype, argtypes, _) in self.protos.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = restype
            fn.argtypes = argtypes
        if missing:
            raise RuntimeError("mantaflow_amd: %s lacks ABI symbols: %s" % (path, ", ".join(missing)))
        self.backend = self.cdll.mf_backend().decode()
        want = int(re.search(r"#define\s+MF_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
        got = int(self.cdll.mf_abi_version())
        if got != want:
            raise RuntimeError("mantaflow_amd: %s implements ABI revision %d, include/manta_hip.h declares %d -- rebuild the library"
                               % (path, got, want))
        # the z-slab window is thread-local state of the shared object (which stays loaded across Library instances): start
        # from "the grid is the whole domain"; solvers carry their own window and set it per call (core.SolverLib)
        self.cdll.mf_set_slab_window(0, 0)
        self.cdll.mf_set_slab_window_source(0, 0)

    def call(self, name, *args):
        fn = getattr(self.cdll, name)
        rc = fn(*args)
        if rc != 0:
            msg = self.cdll.mf_last_error().decode(errors="replace")
            raise RuntimeError(msg)
        return rc


_current = None


def get():
    """The active library (loads the HIP product library on first use; fails loudly if absent)."""
    global _current
    if _current is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("mantaflow_amd: no GPU visible (torch.cuda.is_available() is False); the hot path "
                               "has no CPU fallback")
        _current = Library(DEFAULT_LIB, "cuda")
    return _current


def use_library(path, device):
    """Test hook: route the host layer through another implementation of the same ABI (e.g. the oracle)."""
    global _current
    _current = Library(path, device)
    return _current


def reset():
    global _current
    _current = None
