"""ctypes binding of libcsx.so (include/csx.h).  Thin on purpose: prototypes,
status -> exception mapping, numpy <-> pointer helpers.  There is no CPU
fallback: if the HIP library or a GPU is missing, calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSX_LIB: load another build of the library (the -DCSX_ABLATION build used for the timing experiments)
LIB_PATH = os.environ.get("CSX_LIB") or os.path.join(_HERE, "libcsx.so")

OK, EINVAL, EZEROPIVOT, ENOTSPD, ERUNTIME = 0, 1, 2, 3, 4
TRI_L, TRI_LT, TRI_U, TRI_UT = 0, 1, 2, 3
GAXPY_AUTO, GAXPY_EXACT, GAXPY_WAVE, GAXPY_TILED, GAXPY_ATOMIC = 0, 1, 2, 3, 4

H = C.c_uint64
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

_PROTOS = {
    "csx_init": [C.c_int],
    "csx_finalize": [],
    "csx_sync": [],
    "csx_set_stream": [_vp],
    "csx_device_info": [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "csx_compress": [C.c_int32, C.c_int32, C.c_int64, _i32p, _i32p, _f64p, C.POINTER(H)],
    "csx_add": [H, H, C.c_double, C.c_double, C.POINTER(H)],
    "csx_dupl": [H, C.POINTER(H)],
    "csx_drop": [H, C.c_int, C.c_double, C.POINTER(H)],
    "csx_permute": [H, _i32p, _i32p, C.c_int, C.POINTER(H)],
    "csx_symperm": [H, _i32p, C.c_int, C.POINTER(H)],
    "csx_schol": [H, _i32p, _i32p],
    "csx_order_nd_host": [C.c_int32, _i32p, _i32p, _i32p],
    "csx_norm1": [H, _f64p],
    "csx_cholsol_set_order": [H, C.c_int],
    "csx_cholsol_growth": [H, _f64p],
    "csx_cholsol_sn_info": [H, _i32p, _i32p, _i32p, _i32p, _f64p],
    "csx_csc_invalidate": [H],
    "csx_set_option": [C.c_char_p, C.c_int],
    "csx_get_option": [C.c_char_p, C.POINTER(C.c_int)],
    "csx_gaxpy_host": [C.c_int32, C.c_int32, _i32p, _i32p, _f64p, _f64p, _f64p],
    "csx_csc_col_block": [H, C.c_int32, C.c_int32, C.POINTER(H)],
    "csx_mem_trim": [],
    "csx_mem_info": [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "csx_timer_start": [],
    "csx_timer_stop": [_f64p],
    "csx_csc_upload": [C.c_int32, C.c_int32, _i32p, _i32p, _f64p, C.POINTER(H)],
    "csx_csc_alloc": [C.c_int32, C.c_int32, C.c_int32, C.c_int, C.POINTER(H)],
    "csx_csc_wrap": [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(H)],
    "csx_csc_info": [H, _i32p, _i32p, _i32p, C.POINTER(C.c_int)],
    "csx_csc_download": [H, _i32p, _i32p, _f64p],
    "csx_csc_ptrs": [H, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "csx_free": [H],
    "csx_vec_alloc": [C.c_int64, C.POINTER(H)],
    "csx_vec_upload": [_f64p, C.c_int64, C.POINTER(H)],
    "csx_vec_wrap": [_vp, C.c_int64, C.POINTER(H)],
    "csx_vec_download": [H, _f64p, C.c_int64],
    "csx_vec_write": [H, _f64p, C.c_int64],
    "csx_vec_fill": [H, C.c_double],
    "csx_vec_copy": [H, H],
    "csx_vec_ptr": [H, C.POINTER(_vp), C.POINTER(C.c_int64)],
    "csx_ivec_upload": [_i32p, C.c_int64, C.POINTER(H)],
    "csx_ivec_download": [H, _i32p, C.c_int64],
    "csx_gaxpy": [H, H, H, C.c_int],
    "csx_gaxpy_prepare": [H, C.c_int],
    "csx_transpose": [H, C.c_int, C.POINTER(H)],
    "csx_cumsum": [H, H, C.c_int64, C.POINTER(C.c_int64)],
    "csx_multiply": [H, H, C.POINTER(H)],
    "csx_tri_analyse": [H, C.c_int, C.POINTER(H)],
    "csx_tri_info": [H, _i32p, _i32p, _i32p],
    "csx_tri_solve": [H, H, C.c_int32],
    "csx_tri_set_order": [H, C.c_int],
    "csx_tri_solve_list": [H, _f64p, C.POINTER(C.c_int)],
    "csx_cholsol_solve_list": [H, _f64p, C.POINTER(C.c_int)],
    "csx_tri_order_info": [H, _i32p, _f64p],
    "csx_tri_components": [H, _i32p],
    "csx_permute_vec": [H, H, H, C.c_int32, C.c_int32, C.c_int],
    "csx_lusol_solve": [H, H, H, H, H, H, C.c_int32, C.POINTER(C.c_int)],
    "csx_schol_host": [C.c_int32, _i32p, _i32p, _i32p, _i32p],
    "csx_counts_host": [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, C.c_int, _i32p],
    "csx_chol": [H, _i32p, _i32p, _i32p, C.POINTER(H)],
    "csx_chol_info": [_i32p, _f64p],
    "csx_cholsol_plan": [H, _i32p, C.POINTER(H)],
    "csx_cholsol_factor": [H, C.c_int, C.POINTER(H), C.POINTER(H)],
    "csx_cholsol_factor_info": [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "csx_cholsol_solve": [H, H, C.c_int32],
    "csx_cholsol_graph_info": [H, _i32p, _f64p],
    "csx_cholsol_info": [H, _i32p, _i32p, _i32p],
    "csx_qr_host": [C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f64p, _i32p, _i32p, _i32p, _i32p, C.c_int32, C.c_int32,
                    _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _f64p],
    "csx_qr_apply_host": [C.c_int32, _i32p, _i32p, _f64p, _f64p, C.c_int, _f64p],
    "csx_lu_blocks": [H, C.c_double, C.POINTER(H), C.POINTER(H), _i32p, C.POINTER(C.c_int)],
    "csx_lu_etree": [H, C.c_double, C.POINTER(H), C.POINTER(H), _i32p, C.POINTER(C.c_int)],
    "csx_updown": [H, C.c_int, C.c_int32, _i32p, _f64p, _i32p, C.POINTER(C.c_int)],
    "csx_spsolve": [H, H, _i32p, C.c_int, C.c_int, C.POINTER(H)],
    "csx_happly": [H, H, H, C.c_int32, C.c_int],
    "csx_sqr_host": [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, C.POINTER(C.c_int32),
                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "csx_qr_blocks": [H, _i32p, _i32p, _i32p, C.c_int32, C.POINTER(H), C.POINTER(H), _f64p, C.POINTER(C.c_int)],
    "csx_gaxpy_plan_info": [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "csx_gaxpy_plan_shape": [H, C.POINTER(C.c_int), C.POINTER(C.c_double)],
    "csx_lu_host": [C.c_int32, _i32p, _i32p, _f64p, C.c_double, C.POINTER(_i32p), C.POINTER(_i32p),
                    C.POINTER(_f64p), C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_f64p), _i32p],
    "csx_comm_unique_id": [C.POINTER(C.c_uint8)],
    "csx_comm_init": [C.c_int, C.c_int, C.POINTER(C.c_uint8)],
    "csx_comm_finalize": [],
    "csx_comm_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "csx_comm_barrier": [],
    "csx_comm_allreduce_host": [_f64p, C.c_int, C.c_int],
    "csx_comm_bcast_host": [_vp, C.c_int64, C.c_int],
    "csx_comm_bcast_csc": [C.POINTER(H), C.c_int],
    "csx_comm_bcast_vec": [H, C.c_int],
    "csx_comm_reduce_scatter_vec": [H, H],
    "csx_comm_allreduce_vec": [H],
    "csx_comm_scatter_blocks": [H, H, C.c_int64, C.c_int],
    "csx_comm_gather_blocks": [H, H, C.c_int64, C.c_int],
    "csx_block_cols": [H, C.c_int64, C.c_int32, C.c_int32, C.c_int32, H, C.c_int],
    "csx_gaxpy_sharded_plan": [H, C.POINTER(H)],
    "csx_gaxpy_sharded_rows": [H, _i32p, _i32p],
    "csx_gaxpy_sharded": [H, H, H, C.c_int],
    "csx_gaxpy_sharded_plan_for": [H, C.c_int, C.POINTER(H)],
    "csx_gaxpy_sharded_piece": [H, C.c_int, H],
    "csx_gaxpy_sharded_buffers": [H, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64)],
    "csx_gaxpy_sharded_sum": [H, C.c_int, C.c_int, H],
    "csx_gen_grand": [C.c_int32, C.c_int32, C.c_uint64, C.POINTER(H)],
    "csx_gen_grand_uniform": [C.c_int32, C.c_int32, C.c_uint64, C.POINTER(H)],
    "csx_gen_gspd": [C.c_int32, C.c_int32, C.c_uint64, C.POINTER(H)],
    "csx_gen_vec": [C.c_int64, C.c_uint64, C.c_double, C.c_double, C.POINTER(H)],
    "csx_gen_rhs": [C.c_int32, C.c_int32, C.c_int32, C.POINTER(H)],
}

_lib = None
_ready = False


class CsxError(RuntimeError):
    pass


class NotPositiveDefinite(ArithmeticError):
    pass


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch bundles its own libamdhip64.so (same soname as /opt/rocm's).
    A process that uses both libcsx and torch on the device must have them on ONE runtime, or device pointers of one
    are foreign to the other.  Loading torch's copy by path BEFORE libcsx makes the dynamic loader satisfy libcsx's
    DT_NEEDED libamdhip64.so.7 with it, whatever order `import torch` and `_csx.load()` then come in.  Nothing is
    initialised and torch is not imported.
    Taken when CSX_SHARE_TORCH_HIP=1, or at WORLD_SIZE > 1 with the gloo stand-in as the transport
    (CSX_COMM_BACKEND=gloo: shard.Comm imports torch there).  The RCCL transport (the default at WORLD_SIZE > 1) has no
    torch in the process: libcsx, librccl and libamdhip64 all come from the ROCm installation, the combination
    tests/test_gpu_comm.py runs.  A torch that is already imported has already loaded its runtime, and libcsx then
    binds to it with no help."""
    import sys
    if "torch" in sys.modules:
        return "torch (already imported)"
    want = os.environ.get("CSX_SHARE_TORCH_HIP")
    if want == "0" or (want is None and (int(os.environ.get("WORLD_SIZE", "1")) <= 1
                                         or os.environ.get("CSX_COMM_BACKEND") != "gloo")):
        return None
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


HIP_RUNTIME = None


def load():
    """dlopen libcsx.so and set prototypes (no GPU needed for this)."""
    global _lib, HIP_RUNTIME
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libcsx.so is not built: run `python csparse.py_amd/build.py` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        HIP_RUNTIME = _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, args in _PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.csx_last_error.restype = C.c_char_p
        lib.csx_last_error.argtypes = []
        lib.csx_host_free.restype = None
        lib.csx_host_free.argtypes = [_vp]
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(list(_PROTOS) + ["csx_last_error", "csx_host_free"])


def check(status, what=""):
    if status == OK:
        return
    if status == EZEROPIVOT:
        raise ZeroDivisionError("float division by zero")
    if status == ENOTSPD:
        raise NotPositiveDefinite(what or "matrix is not positive definite")
    if status == EINVAL:
        raise ValueError("libcsx: bad argument" + (" in " + what if what else ""))
    raise CsxError("libcsx %s: %s" % (what, load().csx_last_error().decode("utf-8", "replace")))


def init(device=None):
    """Bind this process to one GPU (LOCAL_RANK by default) and create the context."""
    global _ready
    lib = load()
    if not _ready:
        if device is None:
            device = int(os.environ.get("CSX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        st = lib.csx_init(int(device))
        if st != OK:
            raise CsxError("csx_init(%d) failed: %s -- the MI355X path has no CPU fallback"
                           % (device, lib.csx_last_error().decode("utf-8", "replace")))
        _ready = True
    return lib


def lib():
    return init()


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pi(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def pd(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def new_handle():
    return H(0)


def free(h):
    if h and _lib is not None and _ready:
        _lib.csx_free(h)


def sync():
    check(lib().csx_sync(), "csx_sync")


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    check(lib().csx_device_info(name, 256, C.byref(cus), C.byref(mem)), "csx_device_info")
    return name.value.decode(), cus.value, mem.value


class option(object):
    """with _csx.option("spgemm.one_pass", 0): ...   -- a kernel-selection override for tests (csx_set_option)"""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        old = C.c_int(0)
        check(lib().csx_get_option(self.name, C.byref(old)), "csx_get_option")
        self.old = old.value                      # the value in force on entry: nested blocks restore correctly
        check(lib().csx_set_option(self.name, self.value), "csx_set_option")
        return self

    def __exit__(self, *exc):
        check(lib().csx_set_option(self.name, self.old), "csx_set_option")
        return False


class Timer(object):
    """HIP-event timer on the library's stream (the stream the kernels run on)."""

    def __enter__(self):
        check(lib().csx_timer_start(), "csx_timer_start")
        self.ms = None
        return self

    def __exit__(self, *exc):
        ms = C.c_double(0.0)
        check(lib().csx_timer_stop(C.byref(ms)), "csx_timer_stop")
        self.ms = ms.value
        return False
