"""ctypes binding of libfasn.so (C ABI declared in include/fasn.h).

The library is the product path: if it is missing or does not export every symbol of the header the
import fails loudly — there is no CPU / eager fallback behind these functions.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfasn.so")

FASN_ABI_VERSION = 6
FASN_BWD_ONE_PASS = 1
FASN_DTYPE_F16, FASN_DTYPE_BF16, FASN_DTYPE_F32 = 0, 1, 2
FASN_BIAS_NONE, FASN_BIAS_SAME, FASN_BIAS_F32 = 0, 1, 2
FASN_PATH_NAMES = {0: "plain", 1: "key-padding", 2: "vector mask/bias", 3: "vector bias + key-padding", 4: "element-load (slow)", 5: "fp32"}
FASN_PATH_ELEMENT = 4
FASN_PLAN_FWD, FASN_PLAN_BWD, FASN_PLAN_FWD_WS = 0, 1, 2

# every entry point include/fasn.h declares (tests check the .so exports all of them)
EXPORTS = (
    "fasn_abi_version", "fasn_strerror", "fasn_supported", "fasn_fwd", "fasn_fwd_path", "fasn_bwd_path", "fasn_fwd_workspace_bytes", "fasn_fwd_ws",
    "fasn_bwd_workspace_bytes", "fasn_bwd", "fasn_rng_advance", "fasn_launch_plan",
    "fasn_softmax_n_fwd", "fasn_softmax_n_bwd", "fasn_moments",
)


class View4(Structure):
    _fields_ = [("ptr", c_void_p), ("stride", c_int64 * 4)]


class FwdArgs(Structure):
    _fields_ = [
        ("q", View4), ("k", View4), ("v", View4), ("o", View4),
        ("lse", c_void_p),
        ("mask", View4), ("bias", View4),
        ("bias_dtype", c_int32), ("dtype", c_int32),
        ("B", c_int32), ("H", c_int32), ("Sq", c_int32), ("Sk", c_int32), ("D", c_int32), ("Dv", c_int32),
        ("scale", c_float), ("softmax_n", c_float), ("causal", c_int32), ("dropout_p", c_float),
        ("seed", c_uint64), ("offset", c_uint64),
        ("kv_group", c_int32),
        ("rng_state", c_void_p),
    ]


class BwdArgs(Structure):
    _fields_ = [
        ("fwd", FwdArgs),
        ("dout", View4), ("dq", View4), ("dk", View4), ("dv", View4),
        ("delta", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("dbias", View4),
        ("flags", c_int32),
        ("dbias_dtype", c_int32),
    ]


class FasnError(RuntimeError):
    pass


_lib = None


def load():
    """Load libfasn.so once; raise ImportError with the build recipe if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {os.path.join(_HERE, 'csrc')}`) — there is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}; rebuild it from csrc/")
    lib.fasn_abi_version.restype = c_int32
    lib.fasn_strerror.restype = c_char_p
    lib.fasn_strerror.argtypes = [c_int32]
    lib.fasn_supported.restype = c_int32
    lib.fasn_supported.argtypes = [c_int32, c_int32, c_int32]
    lib.fasn_fwd.restype = c_int32
    lib.fasn_fwd.argtypes = [POINTER(FwdArgs), c_void_p]
    lib.fasn_fwd_path.restype = c_int32
    lib.fasn_fwd_path.argtypes = [POINTER(FwdArgs)]
    lib.fasn_bwd_path.restype = c_int32
    lib.fasn_bwd_path.argtypes = [POINTER(BwdArgs)]
    lib.fasn_fwd_workspace_bytes.restype = c_size_t
    lib.fasn_fwd_workspace_bytes.argtypes = [POINTER(FwdArgs)]
    lib.fasn_fwd_ws.restype = c_int32
    lib.fasn_fwd_ws.argtypes = [POINTER(FwdArgs), c_void_p, c_size_t, c_void_p]
    lib.fasn_bwd.restype = c_int32
    lib.fasn_bwd.argtypes = [POINTER(BwdArgs), c_void_p]
    lib.fasn_rng_advance.restype = c_int32
    lib.fasn_rng_advance.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
    lib.fasn_bwd_workspace_bytes.restype = c_size_t
    lib.fasn_bwd_workspace_bytes.argtypes = [POINTER(BwdArgs)]
    lib.fasn_launch_plan.restype = c_int32
    lib.fasn_launch_plan.argtypes = [POINTER(BwdArgs), c_int32, c_char_p, c_size_t]
    lib.fasn_softmax_n_fwd.restype = c_int32
    lib.fasn_softmax_n_fwd.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_float, c_int32, c_void_p]
    lib.fasn_softmax_n_bwd.restype = c_int32
    lib.fasn_softmax_n_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int32, c_void_p]
    lib.fasn_moments.restype = c_int32
    lib.fasn_moments.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p]
    ver = lib.fasn_abi_version()
    if ver != FASN_ABI_VERSION:
        raise ImportError(f"libfasn ABI version {ver} != expected {FASN_ABI_VERSION}; rebuild csrc/")
    _lib = lib
    return lib


def launch_plan(args, which):
    """The kernels fasn_fwd (FASN_PLAN_FWD), fasn_bwd (FASN_PLAN_BWD) or fasn_fwd_ws (FASN_PLAN_FWD_WS) would launch for `args` (a BwdArgs;
    the forward plans read only its .fwd part): a list of (kernel name with template arguments, grid, block, lds bytes). Nothing is
    launched and no device is touched (include/fasn.h: fasn_launch_plan)."""
    buf = ctypes.create_string_buffer(8192)
    rc = load().fasn_launch_plan(args, which, buf, len(buf))
    if rc < 0:
        check(rc, "fasn_launch_plan")
    out = []
    for line in buf.value.decode().splitlines():
        name, g, b, l, _cfg = line.rsplit(" ", 4)
        out.append((name, int(g.split("=")[1]), int(b.split("=")[1]), int(l.split("=")[1])))
    return out


def launch_plan_described(args, which):
    """The same plan as (kernel family<named template arguments>, grid, block, lds bytes): the `cfg` field of the plan lines (ABI 6) names the
    positional template arguments - fasn_fwd_kernel<bf16,D=64,QB=2,plain,OCC=2,NW=4,RING=2,SEED=2> instead of
    fasn_fwd_kernel<fasn::bf16_tag, 64, 2, 0, 2, 4, 0, 0, 2, 0, 2, 1, 0, 0>. What bench.py prints as roofline.kernels."""
    buf = ctypes.create_string_buffer(8192)
    rc = load().fasn_launch_plan(args, which, buf, len(buf))
    if rc < 0:
        check(rc, "fasn_launch_plan")
    out = []
    for line in buf.value.decode().splitlines():
        name, g, b, l, cfg = line.rsplit(" ", 4)
        cfg = cfg.split("=", 1)[1]
        shown = name if cfg == "-" else f"{name.split('<', 1)[0]}<{cfg}>"
        out.append((shown, int(g.split("=")[1]), int(b.split("=")[1]), int(l.split("=")[1])))
    return out


def check(rc, what):
    if rc != 0:
        msg = load().fasn_strerror(rc).decode()
        raise FasnError(f"{what} failed: {msg} (code {rc})")
