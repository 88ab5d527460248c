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

    # The frontend's vocabulary (the same names oracle/backend_numpy.py implements) onto the entry points of _lib.Context:
    # most names are the Context's own and are forwarded as they are; these differ.
    _RENAMED = {
        "mesh": "mesh_upload", "vec_zeros": "vec_alloc", "vec_to_host": "vec_download", "atom": "atom_assemble",
        "atom_values": "atom_download", "combine": "op_combine", "pcg": "pcg_solve", "slots_get": "slots_download",
        "slots_set": "slots_upload",
    }

    def __getattr__(self, name):
        # (only reached for names the instance does not define itself)
        if name.startswith("_") or name == "ctx":
            raise AttributeError(name)
        fn = getattr(self.ctx, self._RENAMED.get(name, name))
        setattr(self, name, fn)          # bound once: later calls go straight to the Context's method
        return fn

    # ---- device memory as torch tensors (zero copy) for the collectives of pgdrome_amd/dist.py
    def slots_tensor(self):
        """The device scalar bank as a torch tensor for RCCL all-reduces."""
        return _as_torch(self.ctx.slots_ptr(), _lib.NSLOTS, self.device)

    def vec_tensor(self, v):
        return _as_torch(self.ctx.vec_ptr(v), self.ctx.vec_size(v), self.device)


class _CudaArray:
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
        }


def _as_torch(ptr, n, device):
    import torch
    return torch.as_tensor(_CudaArray(ptr, n), device=torch.device("cuda", device))
