"""ctypes binding of include/pgd_amd.h (libpgd_amd.so).

This is the whole Python <-> native boundary: plain pointers, sizes and 64-bit
handles.  Loading never falls back to a CPU implementation: a missing library
or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpgd_amd.so"

H = C.c_int64
I64 = C.c_int64
I32 = C.c_int32
F64 = C.c_double
PD = C.POINTER(C.c_double)
PI32 = C.POINTER(C.c_int32)
PI64 = C.POINTER(C.c_int64)
PH = C.POINTER(C.c_int64)
VP = C.c_void_p

PU8 = C.POINTER(C.c_uint8)
HALO_FN = C.CFUNCTYPE(C.c_int, VP, H, I64, I64, I64, I64)      # pgd_halo_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, VP, C.c_int, C.c_int)      # pgd_allreduce_fn

# name -> (restype, argtypes); mirrors include/pgd_amd.h declaration by declaration
SIGNATURES = {
    "pgd_ctx_create": (C.c_int, [C.c_int, VP, PH]),
    "pgd_ctx_destroy": (C.c_int, [H]),
    "pgd_sync": (C.c_int, [H]),
    "pgd_last_error": (C.c_char_p, [H]),
    "pgd_version": (C.c_int, []),
    "pgd_device_count": (C.c_int, []),
    "pgd_mesh_upload": (C.c_int, [H, PD, I64, C.c_int, PI32, I64, C.c_int, PH]),
    "pgd_mesh_blocked": (C.c_int, [H, H, C.c_int, PH]),
    "pgd_atom_embed": (C.c_int, [H, H, H, C.c_int, C.c_int, F64, H, PH]),
    "pgd_mesh_info": (C.c_int, [H, H, PI64, PI64, PI64, PI32, PI32, PI32]),
    "pgd_mesh_pattern_download": (C.c_int, [H, H, PI32, PI32]),
    "pgd_mesh_sym_info": (C.c_int, [H, H, PI32, PI32, PI32]),
    "pgd_mesh_dict_count": (C.c_int, [H, H, PI32]),
    "pgd_mesh_lattice": (C.c_int, [H, H, PI32, PD]),
    "pgd_mesh_free": (C.c_int, [H, H]),
    "pgd_vec_alloc": (C.c_int, [H, I64, PH]),
    "pgd_vec_free": (C.c_int, [H, H]),
    "pgd_vec_size": (C.c_int, [H, H, PI64]),
    "pgd_vec_upload": (C.c_int, [H, H, PD, I64, I64]),
    "pgd_vec_download": (C.c_int, [H, H, PD, I64, I64]),
    "pgd_vec_ptr": (C.c_int, [H, H, C.POINTER(VP)]),
    "pgd_vec_fill": (C.c_int, [H, H, F64]),
    "pgd_vec_copy": (C.c_int, [H, H, H]),
    "pgd_vec_scale": (C.c_int, [H, H, F64]),
    "pgd_vec_mul": (C.c_int, [H, H, H, H]),
    "pgd_vec_axpy": (C.c_int, [H, H, F64, H]),
    "pgd_vec_set": (C.c_int, [H, H, PI32, PD, I64]),
    "pgd_vec_lincomb": (C.c_int, [H, H, PH, PD, C.c_int]),
    "pgd_vec_dot": (C.c_int, [H, H, H, I64, I64, PD]),
    "pgd_atom_assemble": (C.c_int, [H, H, C.c_int, C.c_int, C.c_int, H, PH]),
    "pgd_atom_upload": (C.c_int, [H, H, PD, PH]),
    "pgd_atom_download": (C.c_int, [H, H, PD]),
    "pgd_atom_free": (C.c_int, [H, H]),
    "pgd_op_combine": (C.c_int, [H, H, PH, PD, C.c_int, PI32, I64, PH]),
    "pgd_spmv": (C.c_int, [H, H, H, H, I64, I64]),
    "pgd_bilinear": (C.c_int, [H, H, H, H, I64, I64, PD]),
    "pgd_bilinear_many": (C.c_int, [H, H, H, PH, C.c_int, I64, I64, PD]),
    "pgd_start_gram": (C.c_int, [H, H, PH, C.c_int, H, I64, I64, PD]),
    "pgd_start_residual": (C.c_int, [H, H, C.c_int, PD, H, H]),
    "pgd_mg_slab_setup": (C.c_int, [H, H, C.c_int, C.c_int, I64, I64, C.POINTER(I64), C.POINTER(C.c_int)]),
    "pgd_mg_slab_fix_start": (C.c_int, [H, H, H, H, I64, I64]),
    "pgd_mg_slab_down": (C.c_int, [H, H, H]),
    "pgd_mg_slab_restrict": (C.c_int, [H, H, H]),
    "pgd_mg_coarse": (C.c_int, [H, H, H]),
    "pgd_mg_slab_up": (C.c_int, [H, H, H, H, H, C.c_int, PD]),
    "pgd_bicgstab_solve": (C.c_int, [H, H, H, H, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), PD]),
    "pgd_atom_product_form": (C.c_int, [H, H, C.POINTER(C.c_int)]),
    "pgd_vec_multidot": (C.c_int, [H, H, PH, C.c_int, I64, I64, PD]),
    "pgd_vec_multidot_pair": (C.c_int, [H, H, H, PH, C.c_int, I64, I64, PD]),
    "pgd_pcg_solve": (C.c_int, [H, H, H, H, F64, F64, C.c_int, C.POINTER(C.c_int), PD]),
    "pgd_band_solve": (C.c_int, [H, H, H, H]),
    "pgd_slots_ptr": (C.c_int, [H, C.POINTER(VP)]),
    "pgd_slots_download": (C.c_int, [H, PD, C.c_int, C.c_int]),
    "pgd_slots_upload": (C.c_int, [H, PD, C.c_int, C.c_int]),
    "pgd_flags_reset": (C.c_int, [H]),
    "pgd_flags_download": (C.c_int, [H, PI32, PI32, PI32]),
    "pgd_op_diag_inv": (C.c_int, [H, H, H]),
    "pgd_spmv_dot_slot": (C.c_int, [H, H, H, H, H, I64, I64, C.c_int]),
    "pgd_pcg_init_slot": (C.c_int, [H, H, H, H, H, H, H, I64, I64, C.c_int]),
    "pgd_pcg_tol_slot": (C.c_int, [H, F64, F64, C.c_int, C.c_int, C.c_int]),
    "pgd_pcg_xr_slot": (C.c_int, [H, H, H, H, H, H, H, I64, I64, C.c_int, C.c_int, C.c_int]),
    "pgd_pcg_check_slot": (C.c_int, [H, C.c_int, C.c_int]),
    "pgd_pcg_p_slot": (C.c_int, [H, H, H, I64, I64, C.c_int, C.c_int]),
    "pgd_cg_init_slot": (C.c_int, [H, H, H, H, H, H, H, H, I64, I64, C.c_int]),
    "pgd_cg_update_slot": (C.c_int, [H, H, H, H, H, H, H, H, I64, I64, C.c_int]),
    "pgd_cg_scalars_slot": (C.c_int, [H, C.c_int, C.c_int, F64, F64]),
    "pgd_comm_bind_callbacks": (C.c_int, [H, HALO_FN, ALLREDUCE_FN, VP, C.c_int, C.c_int]),
    "pgd_comm_unique_id": (C.c_int, [H, PU8]),
    "pgd_comm_bind_rccl": (C.c_int, [H, PU8, C.c_int, C.c_int]),
    "pgd_comm_overlap": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
    "pgd_comm_unbind": (C.c_int, [H]),
    "pgd_comm_timeout": (C.c_int, [H, F64]),
    "pgd_comm_prof": (C.c_int, [H, C.c_int, PD]),
    "pgd_comm_info": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pgd_comm_halo": (C.c_int, [H, H, I64, I64, I64, I64]),
    "pgd_comm_allreduce_slots": (C.c_int, [H, C.c_int, C.c_int]),
    "pgd_comm_push_export": (C.c_int, [H, I64, I64, I64, I64, I64, PU8]),
    "pgd_comm_push_attach": (C.c_int, [H, PU8, PU8, C.POINTER(C.c_int)]),
    "pgd_comm_push": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
    "pgd_comm_allreduce_attach": (C.c_int, [H, PU8, C.POINTER(C.c_int)]),
    "pgd_comm_allreduce_direct": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
    "pgd_pcg_solve_sharded": (C.c_int, [H, H, H, H, I64, I64, I64, I64, F64, F64, C.c_int, C.POINTER(C.c_int), PD]),
    "pgd_op_symmetrize": (C.c_int, [H, H, C.POINTER(C.c_int)]),
    "pgd_op_classify": (C.c_int, [H, H, C.POINTER(C.c_int)]),
    "pgd_tune": (C.c_int, [H, C.c_int, I64]),
    "pgd_prof_enable": (C.c_int, [H, C.c_int]),
    "pgd_prof_read": (C.c_int, [H, PI64, PD, PD]),
    "pgd_prof_read_own": (C.c_int, [H, PD]),
    "pgd_prof_read_update": (C.c_int, [H, PI64, PD, PD]),
    "pgd_prof_read_dropped": (C.c_int, [H, PI64]),
    "pgd_prof_event_overhead": (C.c_int, [H, PD]),
    "pgd_kernel_counts": (C.c_int, [H, PI64, C.c_int]),
    "pgd_classify_counts": (C.c_int, [H, PI64, PI64]),
    "pgd_mg_counts": (C.c_int, [H, PI64, PI64]),
    "pgd_calib_stream": (C.c_int, [H, H, C.c_int, C.c_int]),
    "pgd_timer_start": (C.c_int, [H]),
    "pgd_timer_stop": (C.c_int, [H, PD]),
}

NSLOTS = 64


ERR_TIMEOUT, ERR_PEER = -7, -8          # PGD_ERR_TIMEOUT, PGD_ERR_PEER (include/pgd_amd.h)


class PgdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpgd_amd error {code}: {msg}")
        self.code = code


_lib = None


def load(path: Path | None = None):
    """dlopen libpgd_amd.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: build the HIP library first (python -m pgdrome_amd.build). "
            "pgdrome_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so; if libpgd_amd.so pulls in
    the system copy first, a later `import torch` (needed for the RCCL path) finds the device taken and
    reports "No HIP GPUs are available".  Loading torch's copy first (without importing torch) makes both
    resolve to the same runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(PD)


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(PI32)


class Context:
    """One device context; thin, checked wrappers around the C entry points."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        if self.lib.pgd_device_count() <= 0:
            raise RuntimeError("pgdrome_amd: no HIP device visible (no CPU fallback exists)")
        h = H(0)
        rc = self.lib.pgd_ctx_create(device, VP(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise PgdError(rc, self.lib.pgd_last_error(0).decode())
        self.h = h.value
        self.device = device

    def close(self):
        if getattr(self, "h", 0):
            self.lib.pgd_ctx_destroy(self.h)
            self.h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PgdError(rc, self.lib.pgd_last_error(self.h).decode())

    def sync(self):
        self._ck(self.lib.pgd_sync(self.h))

    # ---- meshes
    def mesh_upload(self, coords, cells):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        if coords.ndim == 1:
            coords = coords.reshape(-1, 1)
        m = H(0)
        self._ck(self.lib.pgd_mesh_upload(self.h, dptr(coords), coords.shape[0], coords.shape[1],
                                          iptr(cells), cells.shape[0], cells.shape[1], C.byref(m)))
        return m.value

    def mesh_blocked(self, mesh, ncomp):
        m = H(0)
        self._ck(self.lib.pgd_mesh_blocked(self.h, mesh, int(ncomp), C.byref(m)))
        return m.value

    def atom_embed(self, bmesh, src, cv, cu, coef=1.0, dst=0):
        out = H(0)
        self._ck(self.lib.pgd_atom_embed(self.h, bmesh, src, int(cv), int(cu), float(coef), int(dst), C.byref(out)))
        return out.value

    def mesh_info(self, mesh):
        nv, nc, nnz = I64(), I64(), I64()
        mr, kl, ku = I32(), I32(), I32()
        self._ck(self.lib.pgd_mesh_info(self.h, mesh, C.byref(nv), C.byref(nc), C.byref(nnz),
                                        C.byref(mr), C.byref(kl), C.byref(ku)))
        return dict(nv=nv.value, nc=nc.value, nnz=nnz.value, max_row=mr.value, kl=kl.value, ku=ku.value)

    def mesh_pattern(self, mesh):
        info = self.mesh_info(mesh)
        rp = np.empty(info["nv"] + 1, dtype=np.int32)
        cols = np.empty(max(info["nnz"], 1), dtype=np.int32)
        self._ck(self.lib.pgd_mesh_pattern_download(self.h, mesh, iptr(rp), iptr(cols)))
        return rp, cols[: info["nnz"]]

    def mesh_sym_info(self, mesh):
        w, nx, ny = I32(), I32(), I32()
        self._ck(self.lib.pgd_mesh_sym_info(self.h, mesh, C.byref(w), C.byref(nx), C.byref(ny)))
        return {"slots": w.value, "nx": nx.value, "ny": ny.value}

    def mesh_lattice(self, mesh):
        """(is_lattice, steps[3]) of pgd_mesh_lattice."""
        flag, steps = I32(), np.zeros(3)
        self._ck(self.lib.pgd_mesh_lattice(self.h, mesh, C.byref(flag), dptr(steps)))
        return bool(flag.value), steps

    def mesh_dict_count(self, mesh):
        n = I32()
        self._ck(self.lib.pgd_mesh_dict_count(self.h, mesh, C.byref(n)))
        return n.value

    def mesh_free(self, mesh):
        self._ck(self.lib.pgd_mesh_free(self.h, mesh))

    # ---- vectors
    def vec_alloc(self, n):
        v = H(0)
        self._ck(self.lib.pgd_vec_alloc(self.h, int(n), C.byref(v)))
        return v.value

    def vec_from(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = self.vec_alloc(a.size)
        self.vec_upload(v, a)
        return v

    def vec_free(self, v):
        self._ck(self.lib.pgd_vec_free(self.h, v))

    def vec_size(self, v):
        n = I64()
        self._ck(self.lib.pgd_vec_size(self.h, v, C.byref(n)))
        return n.value

    def vec_upload(self, v, a, offset=0):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._ck(self.lib.pgd_vec_upload(self.h, v, dptr(a), int(offset), a.size))

    def vec_download(self, v, offset=0, count=None):
        if count is None:
            count = self.vec_size(v) - offset
        out = np.empty(count, dtype=np.float64)
        self._ck(self.lib.pgd_vec_download(self.h, v, dptr(out), int(offset), int(count)))
        return out

    def vec_ptr(self, v):
        p = VP()
        self._ck(self.lib.pgd_vec_ptr(self.h, v, C.byref(p)))
        return p.value

    def vec_fill(self, v, a):
        self._ck(self.lib.pgd_vec_fill(self.h, v, float(a)))

    def vec_copy(self, dst, src):
        self._ck(self.lib.pgd_vec_copy(self.h, dst, src))

    def vec_scale(self, v, a):
        self._ck(self.lib.pgd_vec_scale(self.h, v, float(a)))

    def vec_mul(self, y, a, x):
        self._ck(self.lib.pgd_vec_mul(self.h, y, a, x))

    def vec_axpy(self, y, a, x):
        self._ck(self.lib.pgd_vec_axpy(self.h, y, float(a), x))

    def vec_lincomb(self, y, xs, coefs):
        k = len(xs)
        arr = (H * max(k, 1))(*[int(x) for x in xs])
        cf = np.ascontiguousarray(coefs, dtype=np.float64)
        self._ck(self.lib.pgd_vec_lincomb(self.h, y, arr, dptr(cf) if k else None, k))

    def vec_set(self, v, idx, val):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(np.broadcast_to(np.asarray(val, dtype=np.float64), idx.shape))
        self._ck(self.lib.pgd_vec_set(self.h, v, iptr(idx), dptr(val), idx.size))

    def vec_dot(self, x, y, lo=0, hi=-1):
        out = F64()
        self._ck(self.lib.pgd_vec_dot(self.h, x, y, int(lo), int(hi), C.byref(out)))
        return out.value

    def vec_multidot(self, x, ys, lo=0, hi=-1):
        """[x . y for y in ys] over [lo, hi): one host synchronisation for up to 256 vectors."""
        k = len(ys)
        arr = (H * k)(*[int(v) for v in ys])
        out = np.zeros(k, dtype=np.float64)
        self._ck(self.lib.pgd_vec_multidot(self.h, x, arr, k, int(lo), int(hi), dptr(out)))
        return out

    def vec_multidot_pair(self, x0, x1, ys, lo=0, hi=-1):
        """([x0 . y for y in ys], [x1 . y for y in ys]) over [lo, hi): every y read once for both, one host synchronisation."""
        k = len(ys)
        arr = (H * k)(*[int(v) for v in ys])
        out = np.zeros(2 * k, dtype=np.float64)
        self._ck(self.lib.pgd_vec_multidot_pair(self.h, x0, x1, arr, k, int(lo), int(hi), dptr(out)))
        return out[:k], out[k:]

    # ---- atoms / operators
    def atom_product_form(self, atom):
        """0: CSR kernels, 1: z-march over the diagonal form, 2: z-march over the atom's row classes."""
        f = C.c_int(0)
        self._ck(self.lib.pgd_atom_product_form(self.h, atom, C.byref(f)))
        return f.value

    def atom_assemble(self, mesh, kind, da=0, db=0, w=0):
        a = H(0)
        self._ck(self.lib.pgd_atom_assemble(self.h, mesh, int(kind), int(da), int(db), int(w), C.byref(a)))
        return a.value

    def atom_upload(self, mesh, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        a = H(0)
        self._ck(self.lib.pgd_atom_upload(self.h, mesh, dptr(vals), C.byref(a)))
        return a.value

    def atom_download(self, atom, nnz):
        out = np.empty(max(nnz, 1), dtype=np.float64)
        self._ck(self.lib.pgd_atom_download(self.h, atom, dptr(out)))
        return out[:nnz]

    def atom_free(self, a):
        self._ck(self.lib.pgd_atom_free(self.h, a))

    def op_combine(self, mesh, atoms, coefs, bc_dofs=None, op=0):
        n = len(atoms)
        arr = (H * n)(*[int(a) for a in atoms])
        cf = np.ascontiguousarray(coefs, dtype=np.float64)
        bc = np.ascontiguousarray(bc_dofs if bc_dofs is not None else [], dtype=np.int32)
        o = H(int(op))
        self._ck(self.lib.pgd_op_combine(self.h, mesh, arr, dptr(cf), n,
                                         iptr(bc) if bc.size else None, bc.size, C.byref(o)))
        return o.value

    def spmv(self, A, x, y, r0=0, r1=-1):
        self._ck(self.lib.pgd_spmv(self.h, A, x, y, int(r0), int(r1)))

    def bilinear(self, A, x, y, r0=0, r1=-1):
        out = F64()
        self._ck(self.lib.pgd_bilinear(self.h, A, x, y, int(r0), int(r1), C.byref(out)))
        return out.value

    def bilinear_many(self, A, x, ys, r0=0, r1=-1):
        n = len(ys)
        out = np.zeros(max(n, 1), dtype=np.float64)
        arr = (H * max(n, 1))(*[int(v) for v in ys])
        self._ck(self.lib.pgd_bilinear_many(self.h, A, x, arr, n, int(r0), int(r1), dptr(out)))
        return out[:n]

    def start_gram(self, A, vecs, b, r0=0, r1=-1):
        """(G, g): G[i, j] = v_i . (A v_j), g[j] = v_j . b over rows [r0, r1); one host synchronisation."""
        k = len(vecs)
        arr = (H * k)(*[int(v) for v in vecs])
        out = np.zeros(k * k + k, dtype=np.float64)
        self._ck(self.lib.pgd_start_gram(self.h, A, arr, k, b, int(r0), int(r1), dptr(out)))
        return out[:k * k].reshape(k, k).copy(), out[k * k:].copy()

    def start_residual(self, A, coefs, b, r):
        """r = b - sum_j coefs[j] (A v_j) from the products the start_gram call right before has left in the library."""
        cf = np.ascontiguousarray(coefs, dtype=np.float64)
        self._ck(self.lib.pgd_start_residual(self.h, A, int(cf.size), dptr(cf), b, r))

    def pcg_solve(self, op, b, x, rtol=1e-10, atol=0.0, maxit=10000):
        it = C.c_int()
        rel = F64()
        self._ck(self.lib.pgd_pcg_solve(self.h, op, b, x, float(rtol), float(atol), int(maxit),
                                        C.byref(it), C.byref(rel)))
        return it.value, rel.value

    # ---- the V-cycle on a z-slab of a row-sharded lattice (levels >= 1 whole on every rank)
    def mg_slab_setup(self, op, nz_global, z_first, own0, own1):
        """Entries of a level-1 vector where the slab V-cycle applies to this operator, else 0."""
        n1, ok = I64(0), C.c_int(0)
        self._ck(self.lib.pgd_mg_slab_setup(self.h, op, int(nz_global), int(z_first), int(own0), int(own1), C.byref(n1), C.byref(ok)))
        return int(n1.value) if ok.value else 0

    def mg_slab_fix_start(self, op, b, x, own0, own1):
        self._ck(self.lib.pgd_mg_slab_fix_start(self.h, op, b, x, int(own0), int(own1)))

    def mg_slab_down(self, r, t):
        self._ck(self.lib.pgd_mg_slab_down(self.h, r, t))

    def mg_slab_restrict(self, t, b1):
        self._ck(self.lib.pgd_mg_slab_restrict(self.h, t, b1))

    def mg_coarse(self, b1, x1):
        self._ck(self.lib.pgd_mg_coarse(self.h, b1, x1))

    def mg_slab_up(self, r, x1, t, z, slot=-1):
        """z = M r on the owned planes; r . z to the host (slot < 0) or into that scalar slot without a host synchronisation."""
        if slot >= 0:
            self._ck(self.lib.pgd_mg_slab_up(self.h, r, x1, t, z, int(slot), None))
            return None
        out = F64()
        self._ck(self.lib.pgd_mg_slab_up(self.h, r, x1, t, z, -1, C.byref(out)))
        return out.value

    def bicgstab(self, op, b, x, rtol=1e-10, atol=0.0, maxit=10000):
        it = C.c_int()
        rel = F64()
        self._ck(self.lib.pgd_bicgstab_solve(self.h, op, b, x, float(rtol), float(atol), int(maxit), C.byref(it), C.byref(rel)))
        return it.value, rel.value

    def band_solve(self, op, b, x):
        self._ck(self.lib.pgd_band_solve(self.h, op, b, x))

    # ---- distributed PCG pieces
    def slots_ptr(self):
        p = VP()
        self._ck(self.lib.pgd_slots_ptr(self.h, C.byref(p)))
        return p.value

    def slots_download(self, first=0, count=NSLOTS):
        out = np.empty(count, dtype=np.float64)
        self._ck(self.lib.pgd_slots_download(self.h, dptr(out), int(first), int(count)))
        return out

    def slots_upload(self, vals, first=0):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._ck(self.lib.pgd_slots_upload(self.h, dptr(vals), int(first), vals.size))

    def flags_reset(self):
        self._ck(self.lib.pgd_flags_reset(self.h))

    def flags(self):
        d, i, s = I32(), I32(), I32()
        self._ck(self.lib.pgd_flags_download(self.h, C.byref(d), C.byref(i), C.byref(s)))
        return d.value, i.value, s.value

    def op_diag_inv(self, op, dinv):
        self._ck(self.lib.pgd_op_diag_inv(self.h, op, dinv))

    def spmv_dot_slot(self, A, x, y, w, r0, r1, slot):
        self._ck(self.lib.pgd_spmv_dot_slot(self.h, A, x, y, w, int(r0), int(r1), int(slot)))

    def pcg_init_slot(self, b, q, dinv, r, z, p, lo, hi, slot):
        self._ck(self.lib.pgd_pcg_init_slot(self.h, b, q, dinv, r, z, p, int(lo), int(hi), int(slot)))

    def pcg_tol_slot(self, rtol, atol, slot_rr, slot_bb, slot_tol2):
        self._ck(self.lib.pgd_pcg_tol_slot(self.h, float(rtol), float(atol), slot_rr, slot_bb, slot_tol2))

    def pcg_xr_slot(self, x, r, p, q, dinv, z, lo, hi, slot_rz, slot_pq, slot_out):
        self._ck(self.lib.pgd_pcg_xr_slot(self.h, x, r, p, q, dinv, z, int(lo), int(hi),
                                          slot_rz, slot_pq, slot_out))

    def pcg_check_slot(self, slot_rr, slot_tol2):
        self._ck(self.lib.pgd_pcg_check_slot(self.h, slot_rr, slot_tol2))

    def pcg_p_slot(self, p, z, lo, hi, slot_num, slot_den):
        self._ck(self.lib.pgd_pcg_p_slot(self.h, p, z, int(lo), int(hi), slot_num, slot_den))

    def cg_init_slot(self, b, q, dinv, r, u, p, s, lo, hi, base):
        self._ck(self.lib.pgd_cg_init_slot(self.h, b, q, dinv, r, u, p, s, int(lo), int(hi), int(base)))

    def cg_update_slot(self, x, r, u, w, p, s, dinv, lo, hi, base):
        self._ck(self.lib.pgd_cg_update_slot(self.h, x, r, u, w, p, s, dinv, int(lo), int(hi), int(base)))

    def cg_scalars_slot(self, base, init, rtol, atol):
        self._ck(self.lib.pgd_cg_scalars_slot(self.h, int(base), int(init), float(rtol), float(atol)))

    # ---- in-library sharded solve
    def comm_bind_callbacks(self, halo, allreduce, rank, world):
        """halo(vec, own0, own1, lo_ghost, hi_ghost), allreduce(first_slot, count): Python callables.  An
        exception inside a callback is kept and re-raised by the library call that triggered it."""
        self._cb_error = None

        def _halo(user, vec, own0, own1, lo_g, hi_g):
            try:
                halo(vec, own0, own1, lo_g, hi_g)
                return 0
            except BaseException as e:      # noqa: BLE001 - must not propagate through the C frame
                self._cb_error = e
                return -1

        def _allreduce(user, first, count):
            try:
                allreduce(first, count)
                return 0
            except BaseException as e:      # noqa: BLE001
                self._cb_error = e
                return -1
        self._cbs = (HALO_FN(_halo), ALLREDUCE_FN(_allreduce))      # keep the thunks alive
        self._ck(self.lib.pgd_comm_bind_callbacks(self.h, self._cbs[0], self._cbs[1], None, int(rank), int(world)))

    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        self._ck(self.lib.pgd_comm_unique_id(self.h, buf))
        return bytes(buf)

    def comm_bind_rccl(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("RCCL unique id must be 128 bytes")
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self.lib.pgd_comm_bind_rccl(self.h, buf, int(rank), int(world)))

    def comm_overlap(self, mode=-1):
        """mode 1: try to enable the halo/interior overlap (collective), 0: disable, -1: query; returns the state."""
        st = C.c_int(0)
        self._ck(self.lib.pgd_comm_overlap(self.h, int(mode), C.byref(st)))
        return bool(st.value)

    def comm_unbind(self):
        self._ck(self.lib.pgd_comm_unbind(self.h))
        self._cbs = None

    def comm_timeout(self, seconds):
        """Deadline of the host-side waits of the sharded solve (PGD_ERR_TIMEOUT after that long without progress)."""
        self._ck(self.lib.pgd_comm_timeout(self.h, float(seconds)))

    COMM_PHASES = ("samples", "halo_wait", "product_interior", "product_boundary", "local_sums", "allreduce", "update",
                   "host_boundary_wait")

    def comm_prof(self, mode=-1):
        """Phase timing of the sharded loop: mode 1 = on + reset, 0 = off, -1 = read.  Seconds summed over the samples."""
        out = np.zeros(8)
        self._ck(self.lib.pgd_comm_prof(self.h, int(mode), dptr(out)))
        return dict(zip(self.COMM_PHASES, out.tolist()))

    def comm_info(self):
        k, r, w = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.pgd_comm_info(self.h, C.byref(k), C.byref(r), C.byref(w)))
        return {"kind": ("none", "callbacks", "rccl")[k.value], "rank": r.value, "world": w.value}

    def _ck_cb(self, rc):
        err = getattr(self, "_cb_error", None)
        if err is not None:
            self._cb_error = None
            raise err
        self._ck(rc)

    def comm_halo(self, vec, own0, own1, lo_g, hi_g):
        self._ck_cb(self.lib.pgd_comm_halo(self.h, vec, int(own0), int(own1), int(lo_g), int(hi_g)))

    PUSH_BLOB_BYTES = 256

    def comm_push_export(self, n, own0, own1, lo_g, hi_g):
        """Direct halo, step 1: size the sharded loop's work vectors for this partition; returns the blob the neighbours need."""
        buf = (C.c_uint8 * self.PUSH_BLOB_BYTES)()
        self._ck(self.lib.pgd_comm_push_export(self.h, int(n), int(own0), int(own1), int(lo_g), int(hi_g), buf))
        return bytes(buf)

    def comm_push_attach(self, lower, upper):
        """Direct halo, step 2 (collective over the neighbours): the lower / upper neighbour's blob or None.  True if usable here."""
        def arr(b):
            if b is None:
                return None
            if len(b) != self.PUSH_BLOB_BYTES:
                raise ValueError("comm_push_attach: a blob has %d bytes" % self.PUSH_BLOB_BYTES)
            return (C.c_uint8 * self.PUSH_BLOB_BYTES).from_buffer_copy(b)
        st = C.c_int(0)
        self._ck(self.lib.pgd_comm_push_attach(self.h, arr(lower), arr(upper), C.byref(st)))
        return bool(st.value)

    def comm_allreduce_attach(self, blobs):
        """Direct all-reduce (collective over ALL ranks): every rank's export blob in rank order.  True if usable here."""
        data = b"".join(blobs)
        if len(data) != self.PUSH_BLOB_BYTES * len(blobs):
            raise ValueError("comm_allreduce_attach: blobs of %d bytes each" % self.PUSH_BLOB_BYTES)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        st = C.c_int(0)
        self._ck(self.lib.pgd_comm_allreduce_attach(self.h, buf, C.byref(st)))
        return bool(st.value)

    def comm_allreduce_direct(self, mode=-1):
        st = C.c_int(0)
        self._ck(self.lib.pgd_comm_allreduce_direct(self.h, int(mode), C.byref(st)))
        return bool(st.value)

    def comm_push(self, mode=-1):
        st = C.c_int(0)
        self._ck(self.lib.pgd_comm_push(self.h, int(mode), C.byref(st)))
        return bool(st.value)

    def comm_allreduce_slots(self, first, count):
        self._ck_cb(self.lib.pgd_comm_allreduce_slots(self.h, int(first), int(count)))

    def pcg_solve_sharded(self, op, b, x, own0, own1, lo_g, hi_g, rtol, atol, maxit):
        it, rel = C.c_int(), F64()
        self._ck_cb(self.lib.pgd_pcg_solve_sharded(self.h, op, b, x, int(own0), int(own1), int(lo_g), int(hi_g),
                                                   float(rtol), float(atol), int(maxit), C.byref(it), C.byref(rel)))
        return it.value, rel.value

    def op_symmetrize(self, op):
        used = C.c_int(0)
        self._ck(self.lib.pgd_op_symmetrize(self.h, op, C.byref(used)))
        return bool(used.value)

    def op_classify(self, op):
        """Row-class dictionary of the operator's diagonal form: number of classes, 0 = none (pgd_op_classify)."""
        n = C.c_int(0)
        self._ck(self.lib.pgd_op_classify(self.h, op, C.byref(n)))
        return int(n.value)

    def tune(self, knob, value):
        self._ck(self.lib.pgd_tune(self.h, int(knob), int(value)))

    # ---- measuring
    def prof_enable(self, on=True):
        self._ck(self.lib.pgd_prof_enable(self.h, int(on)))

    def prof_read(self):
        n, s, b = I64(), F64(), F64()
        self._ck(self.lib.pgd_prof_read(self.h, C.byref(n), C.byref(s), C.byref(b)))
        own = F64()
        self._ck(self.lib.pgd_prof_read_own(self.h, C.byref(own)))
        un, us, ub = I64(), F64(), F64()
        self._ck(self.lib.pgd_prof_read_update(self.h, C.byref(un), C.byref(us), C.byref(ub)))
        dr, ov = I64(), F64()
        self._ck(self.lib.pgd_prof_read_dropped(self.h, C.byref(dr)))
        self._ck(self.lib.pgd_prof_event_overhead(self.h, C.byref(ov)))
        return dict(launches=n.value, seconds=s.value, bytes=b.value, own_bytes=own.value,
                    update_launches=un.value, update_seconds=us.value, update_bytes=ub.value, dropped_noop_samples=dr.value,
                    event_overhead=ov.value)

    KERNEL_FAMILIES = ("csr", "csr_dict", "sym_rows", "dia_rows", "dia_march", "multi", "diac_march", "stencil_march")

    def kernel_counts(self):
        out = (C.c_int64 * 8)()
        self._ck(self.lib.pgd_kernel_counts(self.h, out, 8))
        return {k: int(out[i]) for i, k in enumerate(self.KERNEL_FAMILIES)}

    def classify_counts(self):
        full, cached = I64(), I64()
        self._ck(self.lib.pgd_classify_counts(self.h, C.byref(full), C.byref(cached)))
        return {"full": full.value, "cached": cached.value}

    def precondition(self, multigrid):
        """Select the preconditioner of the next pcg_solve calls (1: the multigrid V-cycle where the operator allows it,
        0: Jacobi); returns the number of solves the V-cycle has preconditioned so far."""
        self.tune(40, 1 if multigrid else 0)
        return self.mg_stats()["solves"]

    def mg_stats(self):
        a, b = I64(), I64()
        self._ck(self.lib.pgd_mg_counts(self.h, C.byref(a), C.byref(b)))
        return {"solves": a.value, "fallbacks": b.value}

    def calib_stream(self, v, bytes_per_lane, store=False):
        self._ck(self.lib.pgd_calib_stream(self.h, v, int(bytes_per_lane), int(bool(store))))

    def timer_start(self):
        self._ck(self.lib.pgd_timer_start(self.h))

    def timer_stop(self):
        s = F64()
        self._ck(self.lib.pgd_timer_stop(self.h, C.byref(s)))
        return s.value
