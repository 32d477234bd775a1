"""ctypes binding of libdcgp.so (the C-ABI in include/dcgp.h) and the device context.

Host code is NumPy + ctypes only.  There is NO CPU fallback: if the library is not built, or no
MI355X is visible, anything that computes raises.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCGP_LIB", os.path.join(_HERE, "libdcgp.so"))   # DCGP_LIB: A/B builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dcgp.h")

DCGP_OK, ERR_ARG, ERR_HIP, ERR_NOT_PD, ERR_RCCL, ERR_ALLOC = 0, -1, -2, -3, -4, -5


class LibraryMissing(RuntimeError):
    """libdcgp.so is not built (run `python -c "import __graft_entry__ as g; g.build()"`)."""


class DcgpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdcgp error %d: %s" % (code, msg))
        self.code = code


class NotPositiveDefinite(DcgpError):
    """Cholesky hit a non-positive pivot -- the counterpart of the tf.errors.InvalidArgumentError the
    reference catches at conv_gp/experiment.py:45.  ``column`` is the 1-based failing column."""

    def __init__(self, code, msg, column):
        super().__init__(code, msg)
        self.column = column


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
_i, _d, _sz, _u64 = C.c_int, C.c_double, C.c_size_t, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPE)
_SIGS = {
    "dcgp_ctx_create": [_i, C.POINTER(_vp)],
    "dcgp_ctx_destroy": [_vp],
    "dcgp_last_error": [_vp],
    "dcgp_device_count": [_ip],
    "dcgp_malloc": [_vp, _sz, C.POINTER(_vp)],
    "dcgp_free": [_vp, _vp],
    "dcgp_h2d": [_vp, _vp, _vp, _sz],
    "dcgp_d2h": [_vp, _vp, _vp, _sz],
    "dcgp_memset": [_vp, _vp, _i, _sz],
    "dcgp_sync": [_vp],
    "dcgp_workspace_query": [_vp, C.POINTER(_sz), _ip],
    "dcgp_ctx_set_option": [_vp, C.c_char_p, C.c_long],
    "dcgp_ctx_get_option": [_vp, C.c_char_p, C.POINTER(C.c_long)],
    "dcgp_timing_enable": [_vp, _i],
    "dcgp_timing_reset": [_vp],
    "dcgp_timing_query": [_vp, C.c_char_p, _ip, _dp],
    "dcgp_timing_names": [_vp, C.c_char_p, _sz],
    "dcgp_extract_patches": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i],
    "dcgp_kuu_rbf": [_vp, _vp, _i, _i, _d, _d, _d, _vp],
    "dcgp_kuf_patches_rbf": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _i],
    "dcgp_kuu_acos": [_vp, _vp, _i, _i, _d, _d, _d, _d, _vp],
    "dcgp_kuf_patches_acos": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _d, _d, _vp, _i],
    "dcgp_potrf_lower": [_vp, _vp, _i, _ip],
    "dcgp_trtri_lower": [_vp, _vp, _i, _vp],
    "dcgp_conditional": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _ip],
    "dcgp_conv_layer_forward": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _d, _d, _vp, _vp, _i, _i, _vp, _d,
                                _vp, _vp, _vp, _ip],
    "dcgp_convkernel_kzx": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp],
    "dcgp_convkernel_kdiag": [_vp, _vp, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _vp],
    "dcgp_additive_kdiag": [_vp, _i, _i, _d, _vp, _vp],
    "dcgp_svgp_conditional": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _ip],
    "dcgp_gauss_kl": [_vp, _vp, _vp, _vp, _i, _i, _dp, _ip],
    "dcgp_robustmax_varexp": [_vp, _vp, _vp, _vp, _i, _i, _d, _vp],
    "dcgp_robustmax_predict": [_vp, _vp, _vp, _i, _i, _d, _vp],
    "dcgp_reparam": [_vp, _vp, _vp, _vp, _sz, _d, _vp],
    "dcgp_model_create": [_vp, _i, _d, C.POINTER(_vp)],
    "dcgp_model_destroy": [_vp],
    "dcgp_model_add_conv_layer": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp],
    "dcgp_model_set_head": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp],
    "dcgp_model_set_keep_outputs": [_vp, _i],
    "dcgp_model_set_param": [_vp, _i, C.c_char_p, _vp, _sz],
    "dcgp_elbo_forward": [_vp, _vp, _vp, _i, _d, C.POINTER(_vp), _u64, _i, _dp, _ip],
    "dcgp_elbo_forward_enqueue": [_vp, _vp, _vp, _i, _d, C.POINTER(_vp), _u64, _i, C.POINTER(_u64)],
    "dcgp_elbo_forward_collect": [_vp, _u64, _dp, _ip],
    "dcgp_model_propagate": [_vp, _vp, _i, _i, C.POINTER(_vp), _u64, _vp, _vp, _ip],
    "dcgp_elbo_grad": [_vp, _vp, _vp, _i, _d, C.POINTER(_vp), _u64, _i, C.POINTER(_d), _ip],
    "dcgp_model_get_grad": [_vp, _i, C.c_char_p, _vp, C.c_size_t],
    "dcgp_model_adam_step": [_vp, _d, _d, _d, _d, _i],
    "dcgp_model_train_step_adam": [_vp, _vp, _vp, _i, _d, C.POINTER(_vp), _u64, _i, _d, _d, _d, _d, _i, C.POINTER(_d), _ip],
    "dcgp_model_get_param": [_vp, _i, C.c_char_p, _vp, C.c_size_t],
    "dcgp_model_set_grad_shards": [_vp, _i],
    "dcgp_model_set_shard": [_vp, _i, _i],
    "dcgp_model_grad_block": [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_size_t)],
    "dcgp_model_sgd_step": [_vp, _d],
    "dcgp_model_set_grad_exchange": [_vp, _i],
    "dcgp_model_debug_sharded_adam": [_vp, _i, _d, _d, _d, _d, _i],
    "dcgp_shard_range": [C.c_long, _i, _i, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)],
    "dcgp_model_set_trainable": [_vp, _i, C.c_char_p, _i],
    "dcgp_model_set_factor_reuse": [_vp, _i],
    "dcgp_model_chain_skips": [_vp, C.POINTER(_u64)],
    "dcgp_model_natgrad_step": [_vp, _d, _ip],
    "dcgp_model_predict_y": [_vp, _vp, _i, _i, C.POINTER(_vp), _u64, _vp, _vp, _ip],
    "dcgp_model_layer_output": [_vp, _i, _vp, _vp, _vp, _ip, _ip],
    "dcgp_gemm_strided": [_vp, _vp, C.c_long, C.c_long, C.c_long, _vp, C.c_long, C.c_long, C.c_long, _vp, C.c_long, C.c_long,
                          _i, _i, _i, _i, _d, _i, _vp, C.c_long, C.c_long, _vp, C.c_long, C.c_long, _i],
    "dcgp_gemm_strided_ex": [_vp, _vp, C.c_long, C.c_long, C.c_long, _vp, C.c_long, C.c_long, C.c_long, _vp, C.c_long, C.c_long,
                             _i, _i, _i, _i, _d, _i, _vp, C.c_long, _vp, C.c_long, C.c_long, _i],
    "dcgp_kmeans": [_vp, _vp, C.c_long, _i, _i, _vp, _i, _d, _vp, _ip],
    "dcgp_debug_set_fused_trace": [_vp, _vp],
    "dcgp_debug_comm_gate": [_vp, _i, _ip],
    "dcgp_debug_mfma_f64_rate": [_vp, _dp],
    "dcgp_debug_store_rate": [_vp, _i, _i, _i, _dp],
    "dcgp_debug_set_sweep_trace": [_vp, _vp, C.c_long, C.c_char_p],
    "dcgp_comm_unique_id": [C.c_char_p],
    "dcgp_comm_init_rank": [_vp, _i, _i, C.c_char_p],
    "dcgp_comm_destroy": [_vp],
    "dcgp_comm_count": [_vp, _ip],
    "dcgp_allreduce_sum_f64": [_vp, _vp, _i],
}
_RESTYPE = {"dcgp_last_error": C.c_char_p}

_lib = None


def declared_symbols():
    """Every entry point include/dcgp.h declares (used by the CPU test that the library exports them)."""
    with open(HEADER_PATH) as fh:
        return sorted(set(re.findall(r"\b(dcgp_[a-z0-9_]+)\s*\(", fh.read())))


def lib():
    """Load libdcgp.so once; raises LibraryMissing loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing("%s not found -- build it with __graft_entry__.build() "
                                 "(make -C deepcgp_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = L
    return _lib


def device_count():
    n = C.c_int(0)
    lib().dcgp_device_count(C.byref(n))
    return n.value


class DeviceArray:
    """A float64 / int32 array resident in HBM, owned by the Python side (freed with the object)."""

    def __init__(self, ctx, shape, dtype=np.float64):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = _vp()
        ctx._check(lib().dcgp_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            self.ctx._check(lib().dcgp_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def set(self, host):
        host = np.ascontiguousarray(host, self.dtype)
        assert host.size == self.size, (host.shape, self.shape)
        if self.nbytes:
            self.ctx._check(lib().dcgp_h2d(self.ctx.handle, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.handle:
            lib().dcgp_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One ctx <-> one GPU <-> one HIP stream (not thread-safe)."""

    def __init__(self, device=0):
        L = lib()
        if device_count() <= 0:
            raise DcgpError(ERR_HIP, "no HIP device visible: the conv-GP path has no CPU fallback")
        h = _vp()
        rc = L.dcgp_ctx_create(int(device), C.byref(h))
        if rc != DCGP_OK:
            raise DcgpError(rc, "dcgp_ctx_create(device=%d) failed" % device)
        self.handle = h.value
        self.device = int(device)
        self.nranks, self.rank = 1, 0     # RCCL communicator of this ctx (comm_init)

    def _check(self, rc, info=None):
        if rc == DCGP_OK:
            return
        msg = lib().dcgp_last_error(self.handle)
        msg = msg.decode() if msg else ""
        if rc == ERR_NOT_PD:
            raise NotPositiveDefinite(rc, msg, info.value if info is not None else -1)
        raise DcgpError(rc, msg)

    # memory ------------------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def to_device(self, host, dtype=np.float64):
        host = np.ascontiguousarray(host, dtype)
        return DeviceArray(self, host.shape, dtype).set(host)

    def as_device(self, x, dtype=np.float64):
        return x if isinstance(x, DeviceArray) else self.to_device(x, dtype)

    def sync(self):
        self._check(lib().dcgp_sync(self.handle))

    def workspace_bytes(self):
        """(bytes, count) of the device workspaces the library itself holds for this ctx (dcgp_workspace_query)."""
        b, n = _sz(0), C.c_int(0)
        self._check(lib().dcgp_workspace_query(self.handle, C.byref(b), C.byref(n)))
        return b.value, n.value

    def measured_mfma_f64_tflops(self):
        """Sustained v_mfma_f64_16x16x4_f64 rate of this device right now (csrc/peaks.hip; ~85 ms)."""
        v = C.c_double(0.0)
        self._check(lib().dcgp_debug_mfma_f64_rate(self.handle, C.byref(v)))
        return v.value

    def measured_store_gbs(self, N, P, M):
        """GB/s of a pure store sweep over an [M x N*P] matrix in the K_uf sweep's tile pattern (csrc/peaks.hip)."""
        v = C.c_double(0.0)
        self._check(lib().dcgp_debug_store_rate(self.handle, int(N), int(P), int(M), C.byref(v)))
        return v.value

    # A/B switches ------------------------------------------------------------------------------
    def set_option(self, name, value):
        """One of the ctx's A/B / debugging switches (csrc/common.h DcgpOptions).  The environment variable
        DCGP_<NAME> only gives the initial value at ctx creation; the library never reads the environment on the step path."""
        self._check(lib().dcgp_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_long(0)
        self._check(lib().dcgp_ctx_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """``with ctx.options(no_fused_layer=1): ...`` -- switches set for the block, previous values restored behind it."""
        ctx = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_inner, *exc):
                for k, v in self_inner.old.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    # timing ------------------------------------------------------------------------------------
    def timing_enable(self, on=1):
        """0 off, 1 every kernel family, 2 only the roofline kernels, every 7th launch, 3 the roofline kernels, every launch."""
        self._check(lib().dcgp_timing_enable(self.handle, int(on)))

    def timing_reset(self):
        self._check(lib().dcgp_timing_reset(self.handle))

    def timing(self):
        """{kernel family: (launches, total_ms)} measured with HIP events on the ctx stream."""
        buf = C.create_string_buffer(4096)
        self._check(lib().dcgp_timing_names(self.handle, buf, len(buf)))
        out = {}
        for name in filter(None, buf.value.decode().split(";")):
            n, ms = C.c_int(0), C.c_double(0.0)
            self._check(lib().dcgp_timing_query(self.handle, name.encode(), C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    # strided GEMM on device arrays ---------------------------------------------------------------
    def gemm(self, A, a_strides, B, b_strides, C_out, c_rs, c_bs, M, N, K, batch=1, alpha=1.0, accumulate=False):
        """C_b(i, j) (+)= alpha sum_k A_b(i, k) B_b(k, j) with A_b(i, k) = A[b a_bs + i a_rs + k a_cs] etc. (element strides:
        ``a_strides = (a_rs, a_cs, a_bs)``); dcgp_gemm_strided, the MFMA fp64 GEMM of csrc/gemm_gen.hip."""
        self._check(lib().dcgp_gemm_strided(self.handle, A.ptr, a_strides[0], a_strides[1], a_strides[2], B.ptr, b_strides[0],
                                            b_strides[1], b_strides[2], C_out.ptr, c_rs, c_bs, M, N, K, batch, float(alpha),
                                            int(bool(accumulate)), None, 0, 0, None, 0, 0, 0))

    # multi-GPU ---------------------------------------------------------------------------------
    def comm_init(self, nranks, rank, unique_id):
        self._check(lib().dcgp_comm_init_rank(self.handle, int(nranks), int(rank), bytes(unique_id)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_count(self):
        """Ranks RCCL reports for this ctx's communicator (0 without one)."""
        n = C.c_int(0)
        self._check(lib().dcgp_comm_count(self.handle, C.byref(n)))
        return n.value

    def comm_destroy(self):
        self._check(lib().dcgp_comm_destroy(self.handle))
        self.nranks, self.rank = 1, 0

    def allreduce_sum(self, dev_array):
        self._check(lib().dcgp_allreduce_sum_f64(self.handle, dev_array.ptr, dev_array.size))

    def close(self):
        if getattr(self, "handle", None):
            lib().dcgp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # device arrays may outlive the ctx object in GC order; leave teardown to process exit
        pass


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().dcgp_comm_unique_id(buf)
    if rc != DCGP_OK:
        raise DcgpError(rc, "dcgp_comm_unique_id failed (librccl missing?)")
    return buf.raw


_default_ctx = None


def get_context(device=None):
    """Process-wide default context (device from LOCAL_RANK when launched by torch.distributed.run)."""
    global _default_ctx
    if _default_ctx is None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
            if device >= max(device_count(), 1):
                device = 0
        _default_ctx = Context(device)
    return _default_ctx
