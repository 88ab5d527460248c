"""The HIP engine as a backend of the form frontend: a thin adapter from the
frontend's vocabulary onto the C-ABI (pgdrome_amd/_lib.py -> libpgd_amd.so).

Creating it without the built library or without a GPU raises; nothing in this
package computes on the CPU in its place.
"""
from __future__ import annotations

from . import _lib


class HipBackend:
    name = "hip"

    def __init__(self, device: int = 0, stream: int | None = None):
        self.ctx = _lib.Context(device, stream)
        self.device = device
        self.stream = stream          # external HIP stream handle (None: the library's own stream)
        # PGD_TUNE="knob=value,knob=value": pgd_tune() calls on every context (A/B runs of whole solves without code
        # changes; the knobs select kernels and summation groupings, never a different result beyond rounding)
        import os
        for item in filter(None, os.environ.get("PGD_TUNE", "").split(",")):
            try:
                knob, value = (int(t) for t in item.split("="))
            except ValueError:
                raise ValueError("PGD_TUNE: expected knob=value pairs of integers (include/pgd_amd.h: pgd_tune_knob), got %r" % item) from None
            self.ctx.tune(knob, value)

    # ---- meshes
    def mesh(self, coords, cells):
        return self.ctx.mesh_upload(coords, cells)

    def mesh_blocked(self, mh, ncomp):
        return self.ctx.mesh_blocked(mh, ncomp)

    def atom_embed(self, bmh, src, cv, cu, coef=1.0, dst=0):
        return self.ctx.atom_embed(bmh, src, cv, cu, coef, dst)

    def mesh_info(self, mh):
        return self.ctx.mesh_info(mh)

    def mesh_pattern(self, mh):
        return self.ctx.mesh_pattern(mh)

    def mesh_free(self, mh):
        self.ctx.mesh_free(mh)

    # ---- vectors
    def vec_zeros(self, n):
        return self.ctx.vec_alloc(n)

    def vec_from(self, a):
        return self.ctx.vec_from(a)

    def vec_to_host(self, v):
        return self.ctx.vec_download(v)

    def vec_upload(self, v, a):
        self.ctx.vec_upload(v, a)

    def vec_free(self, v):
        self.ctx.vec_free(v)

    def vec_copy(self, dst, src):
        self.ctx.vec_copy(dst, src)

    def vec_scale(self, v, a):
        self.ctx.vec_scale(v, a)

    def vec_axpy(self, y, a, x):
        self.ctx.vec_axpy(y, a, x)

    def vec_fill(self, v, a):
        self.ctx.vec_fill(v, a)

    def vec_lincomb(self, y, xs, coefs):
        self.ctx.vec_lincomb(y, xs, coefs)

    def vec_set(self, v, idx, vals):
        self.ctx.vec_set(v, idx, vals)

    def vec_dot(self, x, y, lo=0, hi=-1):
        return self.ctx.vec_dot(x, y, lo, hi)

    def vec_multidot(self, x, ys, lo=0, hi=-1):
        return self.ctx.vec_multidot(x, ys, lo, hi)

    # ---- atoms and operators
    def atom_product_form(self, atom):
        return self.ctx.atom_product_form(atom)

    def atom(self, mh, kind, da, db, w):
        return self.ctx.atom_assemble(mh, kind, da, db, w)

    def atom_values(self, a, nnz):
        return self.ctx.atom_download(a, nnz)

    def atom_free(self, a):
        self.ctx.atom_free(a)

    def combine(self, mh, atoms, coefs, bc_vertices=None, reuse=0):
        return self.ctx.op_combine(mh, atoms, coefs, bc_vertices, reuse)

    def spmv(self, A, x, y, r0=0, r1=-1):
        self.ctx.spmv(A, x, y, r0, r1)

    def bilinear(self, A, x, y, r0=0, r1=-1):
        return self.ctx.bilinear(A, x, y, r0, r1)

    def bilinear_many(self, A, x, ys, r0=0, r1=-1):
        return self.ctx.bilinear_many(A, x, ys, r0, r1)

    # ---- solvers
    def start_gram(self, op, vecs, b, r0=0, r1=-1):
        return self.ctx.start_gram(op, vecs, b, r0, r1)

    def pcg(self, op, b, x, rtol, atol, maxit):
        return self.ctx.pcg_solve(op, b, x, rtol, atol, maxit)

    def band_solve(self, op, b, x):
        self.ctx.band_solve(op, b, x)

    # ---- pieces of the row-sharded PCG (pgdrome_amd/dist.py)
    def slots_tensor(self):
        """The device scalar bank as a torch tensor (zero copy) for RCCL all-reduces."""
        return _as_torch(self.ctx.slots_ptr(), _lib.NSLOTS, self.device)

    def vec_tensor(self, v):
        return _as_torch(self.ctx.vec_ptr(v), self.ctx.vec_size(v), self.device)

    def slots_get(self, first=0, count=_lib.NSLOTS):
        return self.ctx.slots_download(first, count)

    def flags_reset(self):
        self.ctx.flags_reset()

    def flags(self):
        return self.ctx.flags()

    def op_diag_inv(self, op, dinv):
        self.ctx.op_diag_inv(op, dinv)

    def spmv_dot_slot(self, A, x, y, w, r0, r1, slot):
        self.ctx.spmv_dot_slot(A, x, y, w, r0, r1, slot)

    def pcg_init_slot(self, b, q, dinv, r, z, p, lo, hi, slot):
        self.ctx.pcg_init_slot(b, q, dinv, r, z, p, lo, hi, slot)

    def pcg_tol_slot(self, rtol, atol, s_rr, s_bb, s_tol2):
        self.ctx.pcg_tol_slot(rtol, atol, s_rr, s_bb, s_tol2)

    def pcg_xr_slot(self, x, r, p, q, dinv, z, lo, hi, s_rz, s_pq, s_out):
        self.ctx.pcg_xr_slot(x, r, p, q, dinv, z, lo, hi, s_rz, s_pq, s_out)

    def pcg_check_slot(self, s_rr, s_tol2):
        self.ctx.pcg_check_slot(s_rr, s_tol2)

    def pcg_p_slot(self, p, z, lo, hi, s_num, s_den):
        self.ctx.pcg_p_slot(p, z, lo, hi, s_num, s_den)

    def cg_init_slot(self, b, q, dinv, r, u, p, s, lo, hi, base):
        self.ctx.cg_init_slot(b, q, dinv, r, u, p, s, lo, hi, base)

    def cg_update_slot(self, x, r, u, w, p, s, dinv, lo, hi, base):
        self.ctx.cg_update_slot(x, r, u, w, p, s, dinv, lo, hi, base)

    def cg_scalars_slot(self, base, init, rtol, atol):
        self.ctx.cg_scalars_slot(base, init, rtol, atol)

    def slots_set(self, vals, first=0):
        self.ctx.slots_upload(vals, first)

    # ---- the sharded solve with loop and communication inside the library (csrc/pgd_comm.hip)
    def comm_bind_callbacks(self, halo, allreduce, rank, world):
        self.ctx.comm_bind_callbacks(halo, allreduce, rank, world)

    def comm_unique_id(self):
        return self.ctx.comm_unique_id()

    def comm_bind_rccl(self, unique_id, rank, world):
        self.ctx.comm_bind_rccl(unique_id, rank, world)

    def comm_overlap(self, mode=-1):
        return self.ctx.comm_overlap(mode)

    def comm_unbind(self):
        self.ctx.comm_unbind()

    def comm_info(self):
        return self.ctx.comm_info()

    def comm_halo(self, vec, own0, own1, lo_g, hi_g):
        self.ctx.comm_halo(vec, own0, own1, lo_g, hi_g)

    def comm_allreduce_slots(self, first, count):
        self.ctx.comm_allreduce_slots(first, count)

    def pcg_solve_sharded(self, op, b, x, own0, own1, lo_g, hi_g, rtol, atol, maxit):
        return self.ctx.pcg_solve_sharded(op, b, x, own0, own1, lo_g, hi_g, rtol, atol, maxit)

    def sync(self):
        self.ctx.sync()

    def prof_enable(self, on=True):
        self.ctx.prof_enable(on)

    def prof_read(self):
        return self.ctx.prof_read()


class _CudaArray:
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
        }


def _as_torch(ptr, n, device):
    import torch
    return torch.as_tensor(_CudaArray(ptr, n), device=torch.device("cuda", device))
