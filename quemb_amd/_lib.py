"""ctypes binding of libqemb_hip.so (C ABI: include/qemb_hip.h; device primitives and measurement hooks: include/qemb_hip_ops.h).

The reference binds its one native helper the same way -- ``ctypes`` + caller-allocated numpy buffers
(shared/external/unrestricted_utils.py:142-160).  The library is built in-tree by ``__graft_entry__.build()``
(``make -C quemb_amd/csrc``).  Nothing here falls back to the CPU: a missing library or device raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libqemb_hip.so"

c_dp = C.POINTER(C.c_double)
c_i64 = C.c_int64
c_vp = C.c_void_p


QEMB_ERR_ARG, QEMB_ERR_ALLOC, QEMB_ERR_DEVICE, QEMB_ERR_NOCONV, QEMB_ERR_NUMERIC = -1, -2, -3, -4, -5      # include/qemb_hip.h


class QembError(RuntimeError):
    """Raised when a libqemb_hip call returns a non-zero status (`.status`: the QEMB_ERR_* code)."""

    def __init__(self, msg="", status=None):
        super().__init__(msg)
        self.status = status


class SolverOpts(C.Structure):
    """qemb_solver_opts (include/qemb_hip.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("cc_conv_tol", C.c_double), ("cc_conv_tol_normt", C.c_double), ("cc_max_cycle", C.c_int),
                ("cc_diis_space", C.c_int), ("scf_conv_tol", C.c_double), ("scf_conv_tol_grad", C.c_double),
                ("scf_max_cycle", C.c_int), ("scf_diis_space", C.c_int), ("warm_start", C.c_int), ("verbose", C.c_int),
                ("relax_density", C.c_int), ("lambda_conv_tol", C.c_double), ("lambda_max_cycle", C.c_int),
                ("strict_convergence", C.c_int)]


_lib = None
_initialised_device = None


def _declare(lib):
    def f(name, restype, *argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    I, D, P, V, L = C.c_int, C.c_double, c_vp, c_vp, c_i64
    f("qemb_init", I, I)
    f("qemb_last_error", C.c_char_p)
    f("qemb_backend", C.c_char_p)
    f("qemb_sync", I)
    f("qemb_device_sync", I)
    f("qemb_mem_info", I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
    f("qemb_malloc", I, C.POINTER(c_vp), C.c_size_t)
    f("qemb_free", I, V)
    f("qemb_trim", I)
    f("qemb_trim_all", I)
    f("qemb_h2d", I, V, V, C.c_size_t)
    f("qemb_d2h", I, V, V, C.c_size_t)
    f("qemb_h2d_async", I, V, V, C.c_size_t)
    f("qemb_d2d", I, V, V, C.c_size_t)
    f("qemb_timer_begin", I, I)
    f("qemb_timer_end", I, I)
    f("qemb_timer_read", I, I, C.POINTER(D), C.POINTER(L))
    f("qemb_timer_reset", I, I)
    f("qemb_timer_live_events", I, I)
    f("qemb_op_gemm", I, L, L, L, D, P, L, I, L, P, L, I, L, D, P, L, L, L)
    f("qemb_op_gemm_slab_rows", I, L, L, L, P, L, L, L, P, L, I, P, L, I)
    f("qemb_op_gemm_probe", I, L, L, L, P, L, I, P, L, I, P, L, I, I, C.POINTER(D), C.POINTER(D), C.POINTER(L))
    f("qemb_op_gemm_stamps", I, L, L, L, P, L, P, L, P, L, I, I, C.POINTER(D))
    f("qemb_set_gemm_config", I, I)
    f("qemb_set_gemm_splitk", I, I)
    f("qemb_set_gemm_ksplit", I, I)
    f("qemb_pair_gemm_choice", I, L, L, C.POINTER(I), C.POINTER(I))
    f("qemb_op_ladder_pack_vvvv", I, L, L, P, P, L, P, L)
    f("qemb_op_ladder_pack_tau", I, L, L, P, P, L, P, L)
    f("qemb_op_ladder_scatter_pm", I, L, L, P, L, P, L, P)
    f("qemb_mfma_f64_peak", I, I, I, C.POINTER(D))
    f("qemb_op_copy4", I, C.POINTER(L), P, C.POINTER(L), P, C.POINTER(L), D, D)
    f("qemb_op_outer4", I, C.POINTER(L), P, L, L, P, L, L, P, C.POINTER(L), D, D)
    f("qemb_op_div_denom", I, P, L, L, L, L, P, P, P, P)
    f("qemb_op_dot", I, L, P, P, P)
    f("qemb_op_absmax", I, L, P, P)
    f("qemb_op_gemv_rows", I, L, L, P, L, P, P, D, D)
    f("qemb_op_gemv_rows_batched", I, L, L, L, P, L, L, P, L, P, D, D)
    f("qemb_op_contract_mid", I, L, L, L, P, P, P, L, D, D)
    f("qemb_op_unpack_s4", I, L, P, P)
    f("qemb_op_pack_s4", I, L, P, P)
    f("qemb_op_unpack_s8_to_s4", I, L, P, P)
    f("qemb_op_unpack_tril_rows", I, L, L, P, P)
    f("qemb_op_pack_pair_rows", I, L, L, P, P)
    f("qemb_op_mirror_lower", I, L, P, L)
    f("qemb_op_k_from_pairs", I, L, P, P, P)
    f("qemb_op_jk_from_packed", I, L, P, P, P, P, P)
    f("qemb_op_pack_pm_cols", I, L, L, P, P, L, P, L)
    f("qemb_op_scatter_pm_rows", I, L, L, P, P, P)
    f("qemb_op_pack_w_pm", I, L, P, P, L, P, L)
    f("qemb_op_ladder_scatter_pm2", I, L, L, P, L, P, L, P, P, I, P)
    f("qemb_op_lincomb2", I, L, D, P, D, P, D, P)
    f("qemb_op_small_k_update", I, L, L, L, L, D, P, L, P, L, P, L)
    f("qemb_op_ccsd_ph_layouts", I, L, L, P, P, P, P, P, P, P, P)
    f("qemb_op_ccsd_y_traces", I, L, L, P, P, P)
    f("qemb_op_copy4_two", I, C.POINTER(L), P, C.POINTER(L), P, C.POINTER(L), D, D, P, P, P, D, D)
    f("qemb_op_scatter_pm_rows_add", I, L, L, P, P, P, P)
    f("qemb_op_ccsd_y_traces_add", I, L, L, P, P, P, P)
    f("qemb_op_pack_w_pm_sum", I, L, P, P, P, P, L, P, L)
    f("qemb_op_ccsd_t1_small", I, L, L, P, P, P, P, P)
    f("qemb_op_gemv_rows2", I, L, L, P, L, P, P, L, P, P, D, D)
    f("qemb_op_ccsd_t1_assemble", I, L, L, P, P, P, P, P, P, P, I, L, P, I, L, P)
    f("qemb_op_gemv_rows_two", I, L, L, P, L, P, P, D, D, L, L, P, L, P, P, D, D)
    f("qemb_op_ccsd_y_traces_slabs", I, L, L, P, P, P, P, I, L, D)
    f("qemb_op_ccsd_finish_t2_rings", I, L, L, P, P, P, P, P, P, P, P)
    f("qemb_op_diis_push", I, L, P, P, P, P, I, P, I, P, P)
    f("qemb_op_ccsd_extrapolate_energy", I, L, L, I, P, P, P, P, P, P)
    f("qemb_op_gather_rows", I, L, L, P, P, L, P)
    f("qemb_op_scale_rows", I, L, L, P, P)
    f("qemb_ctx_count", I, I)
    f("qemb_ctx_bind", I, I)
    f("qemb_ctx_partition", I, I)
    f("qemb_ctx_timer_read", I, I, I, C.POINTER(C.c_double), C.POINTER(c_i64), I)
    f("qemb_gemm_flop_count", I, C.POINTER(C.c_double), I)
    f("qemb_tape_cache_counters", I, C.POINTER(L), C.POINTER(L), I)
    f("qemb_alloc_stats", I, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double), I)
    f("qemb_op_unpack_tril_pair_rows", I, L, L, P, P)
    f("qemb_op_extract_pf", I, L, P, L, L, L, L, L, L, L, L, P)
    f("qemb_op_extract_pf_t", I, L, P, L, L, L, L, L, L, L, L, P)
    f("qemb_op_ladder_pack_vvvv_pf", I, L, L, P, P, L, P, L)
    f("qemb_op_pack_tril_rows", I, L, L, P, P)
    f("qemb_op_jacobi_eigh", I, L, P, P, P, C.POINTER(I))
    f("qemb_op_scf_fused_max", I)
    f("qemb_op_jacobi_eigh_in_basis", I, L, P, P, P, P, P, I, P, D, C.POINTER(I))
    f("qemb_op_scf_fock_small", I, L, P, P, P, P, P, P, P)
    f("qemb_op_pack_density_sym", I, L, P, P)
    f("qemb_op_jacobi_svd", I, L, L, P, P, P, P, C.POINTER(I))
    f("qemb_op_cholesky_lower", I, L, P)
    f("qemb_op_tri_inverse_lower", I, L, P, P)
    f("qemb_comm_unique_id", I, P)
    f("qemb_comm_init", I, I, I, P)
    f("qemb_comm_info", I, C.POINTER(I), C.POINTER(I))
    f("qemb_comm_allreduce", I, P, L, I)
    f("qemb_comm_destroy", I)
    # ---- fragment solver
    OP = C.POINTER(SolverOpts)
    IP = C.POINTER(I)
    DP = C.POINTER(D)
    f("qemb_default_opts", None, OP)
    f("qemb_frag_create", I, I, I, C.POINTER(c_vp))
    f("qemb_frag_free", I, V)
    f("qemb_frag_set_eri_s4", I, V, P)
    f("qemb_frag_set_eri_s4_dev", I, V, P)
    f("qemb_frag_set_df_factor", I, V, I, P)
    f("qemb_frag_set_df_factor_dev", I, V, I, P)
    f("qemb_frag_mo_route", I, V, I)
    f("qemb_frag_set_df_only", I, V, I, P)
    f("qemb_frag_set_df_only_dev", I, V, I, V)
    f("qemb_frag_resident_bytes", I, V, C.POINTER(L))
    f("qemb_frag_mo_route_used", I, V, IP, IP)
    f("qemb_frag_get_eri_s4", I, V, P)
    f("qemb_frag_set_energy_data", I, V, P, P, P, D, IP, I)
    f("qemb_frag_jk", I, V, P, P, P)
    f("qemb_frag_solve", I, V, I, P, P, OP, I, P, P, P, P, P, P, P, DP, DP, DP, IP, IP)
    f("qemb_frag_lambda_iters", I, P, C.POINTER(C.c_int))
    f("qemb_frag_solve_batch", I, I, P, IP, P, P, OP, I, P, P, P, P, P, P, P, P, P, P, IP, IP, C.POINTER(L))
    f("qemb_frag_scf", I, V, I, P, P, OP, P, P, P, P, DP, IP, IP)
    f("qemb_frag_cphf", I, V, I, P, P, OP, P, I, P)
    f("qemb_ccsd_solve", I, I, I, I, P, P, P, OP, P, P, D, IP, I, P, P, P, P, P, P, DP, IP)
    f("qemb_frag_prepare_ccsd", I, V, I, P, P, OP)
    f("qemb_frag_ccsd_iterate", I, V, I, DP, DP)
    f("qemb_frag_ccsd_reset", I, V)
    f("qemb_frag_ccsd_export", I, V, C.c_char_p, P, L)
    # ---- ERI transforms / Schmidt
    LP = C.POINTER(L)
    f("qemb_aoeri_upload", I, I, P, I, C.POINTER(c_vp))
    f("qemb_aoeri_free", I, V)
    f("qemb_ao2mo_dense", I, V, P, I, P, V)
    f("qemb_df_create", I, I, P, C.POINTER(c_vp))
    f("qemb_lpq_upload", I, P, I, C.POINTER(c_vp))
    f("qemb_df_create_pbc", I, I, P, C.POINTER(c_vp), IP)
    f("qemb_df_alloc_ints", I, V, I)
    f("qemb_df_add_pw_block", I, V, I, P, P, P, P)
    f("qemb_df_add_rs_block", I, V, I, I, P)
    f("qemb_df_pw_imag_absmax", I, V, C.POINTER(C.c_double))
    f("qemb_df_pw_select", I, V, I)
    f("qemb_df_free", I, V)
    f("qemb_df_set_ints", I, V, I, P, I)
    f("qemb_df_set_ints_semisparse", I, V, I, L, P, P, P, P)
    f("qemb_df_transform", I, V, P, I, P, V)
    f("qemb_df_transform_screened", I, V, P, I, P, D, P, V)
    f("qemb_df_transform_factor", I, V, P, I, V)
    f("qemb_df_transform_screened_factor", I, V, P, I, P, D, V)
    f("qemb_schmidt", I, P, I, I, I, LP, I, D, P, I, IP, IP)
    f("qemb_schmidt_subspace", I, P, I, I, I, LP, I, D, P, I, IP, IP)
    f("qemb_schmidt_svd", I, P, I, LP, I, D, P, I, IP, IP)
    f("qemb_nsocc_guess", I, P, I, I, P, IP, P)
    f("qemb_matmul", I, L, L, L, P, I, P, I, P)
    f("qemb_abs_overlap_prim", I, I, P, P, P, P, L, I, P, P, P)
    return lib


def load(path: os.PathLike | None = None):
    """dlopen libqemb_hip.so (no device call is made)."""
    global _lib
    if _lib is None:
        p = Path(path) if path else LIB_PATH
        if not p.exists():
            raise QembError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C quemb_amd/csrc). quemb_amd has no CPU fallback."
            )
        _lib = _declare(C.CDLL(str(p)))
    return _lib


def declare(cdll):
    """Attach the C-ABI prototypes to an already opened library (used by tests/hostcheck too)."""
    return _declare(cdll)


class ConvergenceWarning(RuntimeWarning):
    """A fragment solve did not converge and `strict_convergence = 0` asked for PySCF's behaviour (warn, carry on)."""


def check(rc: int, what: str = "", lib=None):
    if rc > 0:          # QEMB_WARN_NOCONV: results were returned
        import warnings
        msg = (lib or load()).qemb_last_error().decode(errors="replace")
        warnings.warn(ConvergenceWarning(f"{what or 'libqemb_hip call'}: {msg}"), stacklevel=3)
    elif rc != 0:
        msg = (lib or load()).qemb_last_error().decode(errors="replace")
        raise QembError(f"{what or 'libqemb_hip call'} failed (status {rc}): {msg}", status=int(rc))


def init(device: int | None = None):
    """Select the GPU (LOCAL_RANK by default) and create the library stream."""
    global _initialised_device
    lib = load()
    if device is None:
        if _initialised_device is not None:
            return lib                      # keep the device chosen by an earlier explicit init()
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _initialised_device != device:
        check(lib.qemb_init(int(device)), "qemb_init")
        _initialised_device = device
    return lib


class DeviceBuffer:
    """A caller-owned FP64 device allocation (freed on __del__ / .free())."""

    def __init__(self, nelem: int, lib=None):
        self.lib = lib or init()
        self.n = int(nelem)
        p = c_vp()
        check(self.lib.qemb_malloc(C.byref(p), max(self.n, 1) * 8), "qemb_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray, lib=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = cls(a.size, lib=lib)
        b.upload(a)
        return b

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size <= self.n
        check(self.lib.qemb_h2d(self.ptr, a.ctypes.data, a.size * 8), "qemb_h2d")

    def numpy(self, shape=None) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        check(self.lib.qemb_d2h(out.ctypes.data, self.ptr, self.n * 8), "qemb_d2h")
        return out if shape is None else out[: int(np.prod(shape))].reshape(shape)

    def at(self, offset_elems: int) -> int:
        return self.ptr + 8 * int(offset_elems)

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.qemb_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def i64x4(v):
    return (c_i64 * 4)(*[int(x) for x in v])
