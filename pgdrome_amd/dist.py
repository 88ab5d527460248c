"""Row-sharding of the large spatial dimension across the GPUs of one node.

The reference has no distributed path; this is the
engine's own.  The structured vertex grid of the spatial mesh is cut into
z-slabs, one per rank (one process per GPU).  A rank holds its owned vertex
planes plus ONE ghost plane on either side and all cells between those planes,
so every owned row of every atom is complete locally; all local vectors have
the extended length and live in local (slab) vertex order:

        [ lo ghost plane | owned planes ............ | hi ghost plane ]
          0 .. own0        own0 .. own1                own1 .. n_ext

Exchange steps (the only collectives on the data path):
  * halo: before an operator is applied to a vector, each rank sends its first /
    last owned plane to the lower / upper neighbour and receives their planes
    into its ghost planes - contiguous slices, no pack kernels, point-to-point
    (2 of the 7 xGMI links per GPU; 512 KiB per face at 256^3);
  * fp64 all-reduce of 1-3 scalars for every dot product.
Both go through ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests), on torch's current stream, which is also the stream the HIP
library enqueues on - so kernels and collectives are ordered without host
synchronisation.  The small time/parameter dimensions are replicated.

The PCG recurrence of a sharded solve runs INSIDE the HIP library when the backend is the HIP one
(``pgd_pcg_solve_sharded``, csrc/pgd_comm.hip: iteration loop, halo ncclSend/ncclRecv and the all-reduce
issued from C++ on the context's stream; the library binds to RCCL with a unique id broadcast from rank 0
here, and checks the binding with a ring shift).  The loops in this file are the same recurrence driven
from Python with the same kernels: they serve the numpy oracle backend in the CPU tests, and remain the
fallback transport (torch.distributed) if the in-library binding cannot be established.
"""
from __future__ import annotations

import os

import logging

import numpy as np

from . import fem

LOG = logging.getLogger("pgdrome_amd.dist")
CHECK_EVERY = 16
# slots of the device scalar bank (shared convention with csrc/pgd_pcg.hip)
S_PQ, S_TOL2, S_FINAL_RR, S_INIT, S_PAIR = 2, 5, 6, 20, 16


class TorchComm:
    """Communication + sharded-solve driver on top of torch.distributed."""

    def __init__(self, dist, backend, single_reduction=None, in_library=True):
        import torch
        self.torch, self.dist, self.be = torch, dist, backend
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._check_stream()
        self.in_library = None        # "rccl" / "callbacks" once the library owns the sharded PCG loop
        # more than one rank: the single-reduction recurrence (one all-reduce per iteration, halo
        # overlapped with the interior rows); one rank: the textbook two-reduction form
        self.single_reduction = (dist.get_world_size() > 1) if single_reduction is None else single_reduction
        self._slots = None
        self._work = {}
        self._views = {}
        # "allreduce": every all-reduce issued through this class; "allreduce_host": those whose result the HOST waits for (the
        # functionals of the callbacks, the Gram data of a solve's start - not the slot all-reduces of a PCG loop)
        self.stats = {"halo": 0, "allreduce": 0, "allreduce_host": 0}
        # PGD_SHARDED_DRIVER=python keeps the recurrence in this file (torch.distributed transport) without a code change
        if in_library and getattr(backend, "name", "") == "hip" and os.environ.get("PGD_SHARDED_DRIVER", "library") != "python":
            self.bind_library()

    halo_overlap = False

    def bind_library(self):
        """Hand the sharded PCG loop and its two communication steps to the HIP library.

        RCCL process group: the library opens its own communicator on the context's stream (unique id
        from rank 0, broadcast here); every rank reports whether its binding and the library's ring-shift
        check succeeded, and only if ALL did is the in-library path used - otherwise all ranks keep the
        torch.distributed transport of this class.  gloo with device memory (tests: ranks sharing one
        GPU): the library calls back into this class for the two steps."""
        import warnings
        be, dist = self.be, self.dist
        if dist.get_backend() != "nccl":
            be.comm_bind_callbacks(self._cb_halo, self._cb_allreduce, self.rank, self.world)
            self.in_library = "callbacks"
            return
        box, why = [None], ""
        if self.rank == 0:
            try:
                box[0] = be.comm_unique_id()
            except Exception as e:          # noqa: BLE001 - reported below, every rank must learn of it
                why = str(e)
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        ok = 0.0
        if box[0] is not None:
            try:
                be.comm_bind_rccl(box[0], self.rank, self.world)
                ok = 1.0
            except Exception as e:          # noqa: BLE001
                why = str(e)
        flag = self.torch.tensor([ok], dtype=self.torch.float64, device=self._scalar_device())
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) == 1.0:
            self.in_library = "rccl"
            # halo exchange concurrent with the interior rows' product (a second communicator + stream): opt-in - it measured
            # slower than the exchange in stream order at every slab size - with PGD_HALO_OVERLAP_MIN_ROWS
            # (the rows per rank from which a solve uses it) or PGD_HALO_OVERLAP=1.  Every rank tries (collective inside the
            # library), all ranks must agree, else it stays off everywhere
            if os.environ.get("PGD_HALO_OVERLAP_MIN_ROWS") is None and os.environ.get("PGD_HALO_OVERLAP", "0") != "1":
                self.halo_overlap = False
                return
            ok = 0.0
            try:
                ok = 1.0 if be.comm_overlap(1) else 0.0
            except Exception as e:          # noqa: BLE001
                why = str(e)
            flag = self.torch.tensor([ok], dtype=self.torch.float64, device=self._scalar_device())
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) != 1.0:
                be.comm_overlap(0)
            self.halo_overlap = float(flag.item()) == 1.0
        else:
            be.comm_unbind()
            warnings.warn("in-library RCCL binding not available on every rank (%s): the sharded PCG is driven "
                          "through torch.distributed instead" % (why or "another rank failed"))

    direct_halo = False
    direct_allreduce = False

    direct_probe = None

    def enable_direct_halo(self, n, own0, own1, lo_g, hi_g, use=True, time_binding=True):
        """Direct halo of the in-library loop (opt-in: PGD_HALO_DIRECT=1, or called by hand): the boundary planes of the search
        direction are stored straight into the neighbours' ghost planes through IPC-mapped pointers and the product waits for a
        posted sequence number - no send / receive kernel in the iteration (include/pgd_amd.h, pgd_comm_push_*).  Collective:
        every rank exports for ITS partition, the blobs travel through torch.distributed, neighbours attach and run a checked
        exchange; all ranks must succeed, else it stays off everywhere (and the solves' own vote would keep it off anyway)."""
        if self.in_library is None:
            return False
        be, dist = self.be, self.dist
        blob, why = None, ""
        try:
            blob = be.comm_push_export(n, own0, own1, lo_g, hi_g)
        except Exception as e:              # noqa: BLE001 - every rank must learn of it
            why = str(e)
        blobs = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(blobs, blob)
        else:
            blobs = [blob]
        ok = 0.0
        if all(b is not None for b in blobs):
            try:
                ok = 1.0 if be.comm_push_attach(blobs[self.rank - 1] if lo_g else None, blobs[self.rank + 1] if hi_g else None) else 0.0
            except Exception as e:          # noqa: BLE001
                why = str(e)
        ok = float(self.allreduce_array([1.0 - ok])[0]) == 0.0
        if not ok:
            try:
                be.comm_push(0)
            except Exception:               # noqa: BLE001
                pass
            LOG.warning("direct halo not available on every rank (%s): the exchange stays with the binding", why or "another rank")
        self.direct_halo = ok
        # ... and the all-reduce of the loop's sums through the ranks' mailboxes (every rank maps every rank; PGD_ALLREDUCE_DIRECT=0
        # keeps the binding's): collective, all ranks must succeed
        self.direct_allreduce = False
        if ok and self.world <= 16 and os.environ.get("PGD_ALLREDUCE_DIRECT", "1") != "0":
            ar = 0.0
            try:
                ar = 1.0 if be.comm_allreduce_attach(blobs) else 0.0
            except Exception as e:          # noqa: BLE001
                why = str(e)
            ar_ok = float(self.allreduce_array([1.0 - ar])[0]) == 0.0
            if not ar_ok:
                try:
                    be.comm_allreduce_direct(0)
                except Exception:           # noqa: BLE001
                    pass
                LOG.warning("direct all-reduce not available on every rank (%s): the loop's sums stay with the binding", why or "another rank")
            self.direct_allreduce = ar_ok
        # what the attach steps' checked exchanges (eight rounds each, every ghost entry compared) said on THIS hardware
        self.direct_probe = {"direct_halo_passed_its_checks_on_every_rank": bool(self.direct_halo),
                             "direct_allreduce_passed_its_checks_on_every_rank": bool(self.direct_allreduce)}
        # ... and what one exchange / one all-reduce of five numbers costs either way here: 50 back to back, one synchronisation
        try:
            self.direct_probe["microseconds"] = self._time_exchanges(n, own0, own1, lo_g, hi_g, binding=time_binding)
        except Exception as e:              # noqa: BLE001 - a probe must not end the run
            self.direct_probe["microseconds"] = {"error": str(e)[:200]}
        if not use:
            # PROBE only (PGD_HALO_DIRECT=probe; bench.py at N > 1): the solves keep the binding's exchange and all-reduce
            for switch in (be.comm_push, be.comm_allreduce_direct):
                try:
                    switch(0)
                except Exception:           # noqa: BLE001
                    pass
            self.direct_halo = self.direct_allreduce = False
        return ok

    def _time_exchanges(self, n, own0, own1, lo_g, hi_g, reps=50, binding=True):
        """Microseconds per halo exchange of one vector and per all-reduce of five scalars: through the binding (RCCL, or the
        callbacks) and - where attached - through the direct paths.  Collective: every rank issues the same sequence."""
        import time
        be, out = self.be, {}
        v = be.vec_zeros(n)

        def timed(fn):
            fn()
            be.sync()
            self.dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            be.sync()
            return 1e6 * (time.perf_counter() - t0) / reps
        try:
            if self.in_library and binding:
                out["halo_through_the_binding"] = timed(lambda: be.comm_halo(v, own0, own1, lo_g, hi_g))
                out["allreduce_through_the_binding"] = timed(lambda: be.comm_allreduce_slots(48, 5))
            if self.direct_halo:
                out["direct_halo"] = timed(lambda: be.comm_push(2))
            if self.direct_allreduce:
                out["direct_allreduce"] = timed(lambda: be.comm_allreduce_direct(2))
        finally:
            be.vec_free(v)
        return out

    def _cb_halo(self, vec, own0, own1, lo_g, hi_g):
        from types import SimpleNamespace
        self.halo_exchange_raw(SimpleNamespace(part=SimpleNamespace(own0=own0, own1=own1, lo_ghost=lo_g, hi_ghost=hi_g)), vec)

    def _cb_allreduce(self, first, count):
        self.allreduce_slots(first, count)

    def _check_stream(self):
        """Collectives are ordered against torch's current stream: the HIP library must enqueue on it too."""
        if getattr(self.be, "name", "") != "hip":
            return
        cur = self.torch.cuda.current_stream().cuda_stream
        if not self.be.stream or self.be.stream != cur:
            raise RuntimeError("sharded runs need HipBackend(device, stream) with stream == torch's current "
                               "(non-default) stream; got backend stream %r, torch current %r" % (self.be.stream, cur))

    # ---- scalars
    def _scalar_device(self):
        return self.slots().device

    def slots(self):
        if self._slots is None:
            self._slots = self.be.slots_tensor()
        return self._slots

    def allreduce_sum(self, value):
        dev = "cpu" if self.dist.get_backend() == "gloo" else self._scalar_device()
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t)
        self.stats["allreduce"] += 1
        self.stats["allreduce_host"] += 1
        return float(t.item())

    def allreduce_maxloc(self, value, payload):
        """(max over ranks of value, payload of the rank that holds it; the lowest such rank on ties): the "delta" stop
        test of the fixed-point loop on a row-sharded dimension (solver.py:763-776 is written for one process)."""
        box = [None] * self.world
        self.dist.all_gather_object(box, (float(value), float(payload)))
        self.stats["allreduce"] += 1
        best = max(range(self.world), key=lambda r: (box[r][0], -r))
        return box[best]

    def allreduce_array(self, values):
        """Sum of a small host array over the ranks in ONE collective (Gram data of the Galerkin start, batched functionals)."""
        dev = "cpu" if self.dist.get_backend() == "gloo" else self._scalar_device()
        t = self.torch.tensor(np.asarray(values, dtype=np.float64), dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t)
        self.stats["allreduce"] += 1
        self.stats["allreduce_host"] += 1
        return t.cpu().numpy()

    def _staged(self, t):
        """gloo cannot move device memory: stage through the host (test configuration: several ranks
        sharing one GPU over gloo; production runs use RCCL and never take this path)."""
        return t.is_cuda and self.dist.get_backend() == "gloo"

    def allreduce_slots(self, first, count):
        t = self.slots()[first:first + count]
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)
        self.stats["allreduce"] += 1

    # ---- halo exchange
    def _halo_ops(self, mesh, t):
        part, P2P, d = mesh.part, self.dist.P2POp, self.dist
        ops = []
        # (what goes down / up has the NEIGHBOUR's ghost size: the same as this rank's on P1 slabs, not on P2 ones - fem.Partition)
        send_lo, send_hi = getattr(part, "send_lo", part.lo_ghost), getattr(part, "send_hi", part.hi_ghost)
        if part.lo_ghost:
            ops.append(P2P(d.isend, t[part.own0:part.own0 + send_lo], self.rank - 1))
            ops.append(P2P(d.irecv, t[0:part.lo_ghost], self.rank - 1))
        if part.hi_ghost:
            ops.append(P2P(d.isend, t[part.own1 - send_hi:part.own1], self.rank + 1))
            ops.append(P2P(d.irecv, t[part.own1:part.own1 + part.hi_ghost], self.rank + 1))
        return ops

    def halo_exchange_raw(self, mesh, handle, cache_view=False):
        """Neighbour planes -> ghost planes of a device vector (contiguous slices, grouped send/recv).
        For the PCG work vectors the tensor view and the P2P descriptors are built once and reused."""
        if cache_view:
            ops = self._views.get(handle)
            if ops is None:
                ops = self._halo_ops(mesh, self.be.vec_tensor(handle))
                self._views[handle] = ops
        else:
            ops = self._halo_ops(mesh, self.be.vec_tensor(handle))
        if ops and self._staged(ops[0].tensor):
            host = [(op, op.tensor.cpu()) for op in ops]
            reqs = self.dist.batch_isend_irecv([self.dist.P2POp(op.op, h, op.peer) for op, h in host])
            for req in reqs:
                req.wait()
            for op, h in host:
                if op.op is self.dist.irecv:
                    op.tensor.copy_(h)
        elif ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        self.stats["halo"] += 1

    def halo_exchange(self, mesh, vec):
        """Refresh the ghost planes of a frontend Vector (skipped while it is unchanged)."""
        if getattr(vec, "_halo_version", None) == vec.version:
            return
        h = vec.dev()
        self.halo_exchange_raw(mesh, h)
        vec._host_ok = False           # ghost entries changed on the device only
        vec._halo_version = vec.version

    # ---- sharded Jacobi-PCG
    def _workvec(self, n, name):
        key = (n, name)
        h = self._work.get(key)
        if h is None:
            h = self.be.vec_zeros(n)
            self._work[key] = h
        return h

    def _halo_begin(self, mesh, handle):
        """Start the halo exchange of a PCG work vector; returns a callable that completes it."""
        ops = self._views.get(handle)
        if ops is None:
            ops = self._halo_ops(mesh, self.be.vec_tensor(handle))
            self._views[handle] = ops
        self.stats["halo"] += 1
        if not ops:
            return lambda: None
        if self._staged(ops[0].tensor):
            self.stats["halo"] -= 1
            self.halo_exchange_raw(mesh, handle, cache_view=True)
            return lambda: None
        reqs = self.dist.batch_isend_irecv(ops)
        return lambda: [req.wait() for req in reqs]

    def _spmv_dot_overlapped(self, mesh, op, u, w, base):
        """w = A u on the owned rows with S[base..base+2] <- local (w.u) of the interior / low / high
        boundary rows.  The interior rows do not touch ghost entries, so they run while the planes travel."""
        be, part = self.be, mesh.part
        lo, hi, glo, ghi = part.own0, part.own1, part.lo_ghost, part.hi_ghost
        finish = self._halo_begin(mesh, u)
        if hi - lo < glo + ghi:          # a rank that owns a single plane: nothing to overlap
            glo, ghi = hi - lo, 0
        # all three slots are written every time (an empty range writes 0): they are all-reduced together
        be.spmv_dot_slot(op, u, w, u, lo + glo, hi - ghi, base)
        finish()
        be.spmv_dot_slot(op, u, w, u, lo, lo + glo, base + 1)
        be.spmv_dot_slot(op, u, w, u, hi - ghi, hi, base + 2)

    def pcg_single_reduction(self, mesh, op, b, x, rtol, atol, maxit):
        """Chronopoulos-Gear Jacobi-PCG: the same Krylov iterates as `pcg`, one all-reduce per iteration."""
        be, part = self.be, mesh.part
        self._check_stream()
        lo, hi, n = part.own0, part.own1, mesh.num_vertices()
        r, u, w, p, s, q, dinv = (self._workvec(n, k) for k in ("r", "z", "w", "p", "s", "q", "dinv"))
        xh, bh = x.dev(), b.dev()
        B = 24
        be.flags_reset()
        be.slots_set(np.zeros(9), B)
        be.op_diag_inv(op, dinv)
        self.halo_exchange_raw(mesh, xh)
        be.spmv(op, xh, q, lo, hi)
        be.cg_init_slot(bh, q, dinv, r, u, p, s, lo, hi, B)            # local (r.u, r.r) and b.b
        self._spmv_dot_overlapped(mesh, op, u, w, B + 2)
        self.allreduce_slots(B, 9)                                        # alpha/beta/old r.u are still 0
        be.cg_scalars_slot(B, 1, rtol, atol)
        k = 0
        while True:
            done, iters, status = be.flags()
            if done or k >= maxit:
                break
            for _ in range(min(CHECK_EVERY, maxit - k)):
                be.cg_update_slot(xh, r, u, w, p, s, dinv, lo, hi, B)
                self._spmv_dot_overlapped(mesh, op, u, w, B + 2)
                self.allreduce_slots(B, 5)                                # (r.u, r.r, w.u x 3) in one call
                be.cg_scalars_slot(B, 0, rtol, atol)
                k += 1
        if status != 0:
            raise RuntimeError("sharded PCG breakdown (NaN) after %d iterations" % iters)
        sl = be.slots_get(0, 40)
        bb, rr = sl[B + 8], sl[S_FINAL_RR]
        x.touched_dev()
        self.halo_exchange(mesh, x)
        return iters, (np.sqrt(rr / bb) if bb > 0 else 0.0)

    def _stuck(self, err):
        """The library's deadline expired inside a sharded solve (PGD_ERR_TIMEOUT): a neighbour died or a collective was
        never matched.  This rank's stream is stuck behind that collective for good - tearing the process group down would
        block on it as well - so the PROCESS ends here, non-zero, with the library's one-line diagnosis (rank, iteration, last
        collective issued) on stderr.  PGD_COMM_TIMEOUT_ACTION=raise hands the error to the caller instead (tests)."""
        import sys
        sys.stderr.write("[pgdrome_amd.dist] rank %d/%d: %s\n[pgdrome_amd.dist] collectives so far: %r; exiting with status 3\n"
                         % (self.rank, self.world, err, self.stats))
        sys.stderr.flush()
        if os.environ.get("PGD_COMM_TIMEOUT_ACTION", "exit") == "raise":
            raise err
        os._exit(3)

    def bicgstab(self, mesh, op, b, x, rtol, atol, maxit):
        """BiCGStab with right Jacobi preconditioning for a NON-symmetric operator on a row-sharded mesh (a convection atom on
        the spatial dimension; the reference's MUMPS solves whatever the callbacks produce, solver.py:627-636): the recurrence of
        csrc/pgd_krylov.hip driven from the host over the backend's vector primitives - every product behind a halo exchange of
        its argument, every group of dots one all-reduce, all decisions (stop test, breakdown restarts) taken on all-reduced
        numbers, hence the same on every rank.  Host-synchronised three times per iteration like the unsharded loop.  Ends with the
        TRUE residual; returns (iterations, relative residual); x comes back with current ghost planes."""
        be, part = self.be, mesh.part
        self._check_stream()
        lo, hi, n = part.own0, part.own1, mesh.num_vertices()
        xh, bh = x.dev(), b.dev()
        r, rhat, p, v, y, z, t, dinv = (self._workvec(n, k) for k in ("bi_r", "bi_rhat", "bi_p", "bi_v", "bi_y", "bi_z", "bi_t", "bi_dinv"))
        be.op_diag_inv(op, dinv)

        def sums(pairs):
            """all-reduced dots over the owned rows: [(a, b), ...] -> [a . b, ...]"""
            loc = [be.vec_dot(a, c, lo, hi) for a, c in pairs]
            return [float(t_) for t_ in self.allreduce_array(loc)]

        def product(src, dst):
            self.halo_exchange_raw(mesh, src)
            be.spmv(op, src, dst, lo, hi)

        def residual():
            product(xh, v)
            be.vec_copy(r, bh)
            be.vec_axpy(r, -1.0, v)
            return sums([(r, r), (bh, bh)])

        rr, bb = residual()
        bnorm = float(np.sqrt(bb))
        tol = max(rtol * bnorm, atol)
        if not np.isfinite(rr):
            raise RuntimeError("sharded BiCGStab: the start residual is not finite")
        it, restarts, fresh = 0, 0, True
        rho = alpha = omega = 1.0
        rho_new = rr

        def restart(what):
            nonlocal restarts, fresh, rr
            restarts += 1
            if restarts > 4:
                raise RuntimeError("sharded BiCGStab: breakdown (%s) after %d iterations" % (what, it))
            rr = residual()[0]
            fresh = True

        for leg in range(4):                       # the recurrence, then up to three short legs from the true residual
            it_leg = 0
            while np.sqrt(rr) > tol and it < maxit and (leg == 0 or it_leg < 50):
                if fresh:
                    be.vec_copy(rhat, r)
                    be.vec_fill(p, 0.0)
                    be.vec_fill(v, 0.0)
                    rho = alpha = omega = 1.0
                    rho_new = rr
                    fresh = False
                beta = (rho_new / rho) * (alpha / omega)
                be.vec_axpy(p, -omega, v)          # p = r + beta (p - omega v)
                be.vec_scale(p, beta)
                be.vec_axpy(p, 1.0, r)
                be.vec_mul(y, dinv, p)
                product(y, v)
                hv, = sums([(rhat, v)])
                if hv == 0.0 or not np.isfinite(hv):
                    restart("rhat . v = 0")
                    continue
                alpha = rho_new / hv
                be.vec_axpy(r, -alpha, v)          # r now holds s
                be.vec_mul(z, dinv, r)
                it += 1
                it_leg += 1
                product(z, t)
                ss, ts, tt = sums([(r, r), (t, r), (t, t)])
                if np.sqrt(ss) <= tol:             # converged in the half step
                    be.vec_axpy(xh, alpha, y)
                    rr = ss
                    break
                if tt == 0.0 or not np.isfinite(tt):
                    be.vec_axpy(xh, alpha, y)
                    restart("t . t = 0")
                    continue
                omega = ts / tt
                be.vec_axpy(xh, alpha, y)
                be.vec_axpy(xh, omega, z)
                be.vec_axpy(r, -omega, t)
                rr, hr = sums([(r, r), (rhat, r)])
                rho, rho_new = rho_new, hr
                if not np.isfinite(rr):
                    raise RuntimeError("sharded BiCGStab: the residual is not finite after %d iterations" % it)
                if omega == 0.0 or rho_new == 0.0:
                    if np.sqrt(rr) <= tol:
                        break
                    restart("omega or rho = 0")
            rr = residual()[0]                      # the recurrence residual drifts from the true one: confirm on b - A x
            if np.sqrt(rr) <= tol * 1.0000001 or it >= maxit:
                break
            fresh = True
        self.halo_exchange_raw(mesh, xh)
        return it, (float(np.sqrt(rr)) / bnorm if bnorm > 0.0 else float(np.sqrt(rr)))

    def pcg_mg(self, mesh, op, b, x, rtol, atol, maxit):
        """PCG preconditioned by the V-cycle on a row-sharded lattice (settings["preconditioner"] = "amg" on a sharded mesh):
        level 0 of the hierarchy stays with the rows - every rank runs the level's stencil passes, restriction and prolongation
        on its own z-slab - and levels >= 1 are whole on every rank (pgd_mg_slab_*, csrc/pgd_mg.hip; oracle/mg_numpy.py::Slab).
        Per iteration: halo(p) + the product, and inside the cycle halo(r), halo(t) and ONE all-reduce of the level-1 right-hand
        side (n / 8 doubles, every entry with a single non-zero contribution: exact); two small all-reduces for the scalars.
        The textbook recurrence with the stop test of the Jacobi form, driven from the host (a solve takes ~20 iterations).
        Returns None - on EVERY rank, decided by an all-reduce - where the cycle does not apply to this operator on some rank;
        the caller then takes the Jacobi-PCG."""
        be, part = self.be, mesh.part
        plane = int(getattr(part, "plane", 0) or 0)
        n1 = 0
        if plane and hasattr(be, "mg_slab_setup"):
            try:
                n1 = int(be.mg_slab_setup(op, part.n_global // plane, part.global_offset // plane, part.own0, part.own1))
            except Exception as e:      # noqa: BLE001 - a rank-local failure must still reach the vote below
                LOG.warning("slab V-cycle setup failed on rank %d: %s", self.rank, e)
                n1 = 0
        votes = self.allreduce_array([0.0 if n1 > 0 else 1.0, float(n1)])
        if votes[0] > 0.0 or votes[1] != float(n1) * self.world:
            return None
        self._check_stream()
        lo, hi, n = part.own0, part.own1, mesh.num_vertices()
        r, z, p, q, t = (self._workvec(n, k) for k in ("r", "z", "p", "q", "mg_t"))
        b1, x1 = self._workvec(n1, "mg_b1"), self._workvec(n1, "mg_x1")
        xh, bh = x.dev(), b.dev()
        b1_t = be.vec_tensor(b1)

        def cycle(slot):
            """z = M r (r: owned rows current); the local r . z lands in scalar slot `slot` (no host synchronisation)"""
            self.halo_exchange_raw(mesh, r, cache_view=True)
            be.mg_slab_down(r, t)
            self.halo_exchange_raw(mesh, t, cache_view=True)
            be.mg_slab_restrict(t, b1)
            if self.world > 1:
                if self._staged(b1_t):
                    hst = b1_t.cpu()
                    self.dist.all_reduce(hst)
                    b1_t.copy_(hst)
                else:
                    self.dist.all_reduce(b1_t)
                self.stats["allreduce"] += 1
            be.mg_coarse(b1, x1)
            be.mg_slab_up(r, x1, t, z, slot)

        # The recurrence of `pcg` below with z = M r in place of z = D^-1 r: scalars in the device slot bank, the rank sums
        # all-reduced there, the stop test on the device - ONE host synchronisation per iteration (the look at the flags).
        ones = self._work.get((n, "mg_ones"))
        if ones is None:
            ones = self._workvec(n, "mg_ones")
            be.vec_fill(ones, 1.0)
        be.flags_reset()
        be.mg_slab_fix_start(op, bh, xh, lo, hi)
        self.halo_exchange_raw(mesh, xh)
        be.spmv(op, xh, q, lo, hi)
        be.pcg_init_slot(bh, q, ones, r, z, p, lo, hi, S_INIT)          # r = b - q; local (r.r as r.z, r.r, b.b)
        cycle(S_INIT)                                                    # z = M r, the local r.z over the slot's r.r
        be.vec_copy(p, z)
        self.allreduce_slots(S_INIT, 3)
        be.pcg_tol_slot(rtol, atol, S_INIT + 1, S_INIT + 2, S_TOL2)       # (raises the flag for a start that already meets the bar)
        rz_old, k = S_INIT, 0
        while True:
            done, iters, status = be.flags()
            if done or k >= maxit:
                break
            out = S_PAIR + 2 * (k & 1)
            self.halo_exchange_raw(mesh, p, cache_view=True)
            be.spmv_dot_slot(op, p, q, p, lo, hi, S_PQ)                  # q = A p, local p.q
            self.allreduce_slots(S_PQ, 1)
            be.pcg_xr_slot(xh, r, p, q, ones, z, lo, hi, rz_old, S_PQ, out)      # x += alpha p, r -= alpha q; local (.., r.r) in out, out + 1
            cycle(out)                                                   # z = M r, the local r.z into `out`
            self.allreduce_slots(out, 2)
            be.pcg_check_slot(out + 1, S_TOL2)
            be.pcg_p_slot(p, z, lo, hi, out, rz_old)
            rz_old = out
            k += 1
        if status != 0:
            raise RuntimeError("sharded multigrid PCG breakdown (NaN residual) after %d iterations" % iters)
        sl = be.slots_get(0, 24)
        bb = sl[S_INIT + 2]
        rr = sl[S_FINAL_RR] if iters > 0 else sl[S_INIT + 1]
        self.stats["sharded_mg_solves"] = self.stats.get("sharded_mg_solves", 0) + 1
        x.touched_dev()
        self.halo_exchange(mesh, x)
        return iters, (float(np.sqrt(rr / bb)) if bb > 0 else 0.0)

    def pcg(self, mesh, op, b, x, rtol, atol, maxit):
        # (a partition whose send sizes differ from its ghost sizes - P2 slabs - takes the loop below: its exchanges are sized per
        # direction, and it multiplies only after the exchange; the same on every rank, the geometry being global knowledge)
        symmetric = getattr(mesh.part, "symmetric", True)
        if self.in_library and symmetric:
            part = mesh.part
            self._check_stream()
            try:
                iters, rel = self.be.pcg_solve_sharded(op, b.dev(), x.dev(), part.own0, part.own1, part.lo_ghost,
                                                       part.hi_ghost, rtol, atol, maxit)
            except Exception as e:      # noqa: BLE001 - only the deadline is handled here
                if getattr(e, "code", 0) == -7:
                    self._stuck(e)
                raise
            self.stats["sharded_solves"] = self.stats.get("sharded_solves", 0) + 1
            x.touched_dev()
            x._host_ok = False
            x._halo_version = x.version      # the library returned x with current ghost planes
            return iters, rel
        if self.single_reduction and symmetric:
            return self.pcg_single_reduction(mesh, op, b, x, rtol, atol, maxit)
        be, part = self.be, mesh.part
        self._check_stream()
        lo, hi, n = part.own0, part.own1, mesh.num_vertices()
        r, z, p, q, dinv = (self._workvec(n, k) for k in ("r", "z", "p", "q", "dinv"))
        xh, bh = x.dev(), b.dev()
        be.flags_reset()
        be.op_diag_inv(op, dinv)
        self.halo_exchange_raw(mesh, xh)
        be.spmv(op, xh, q, lo, hi)
        be.pcg_init_slot(bh, q, dinv, r, z, p, lo, hi, S_INIT)          # local (r.z, r.r, b.b)
        self.allreduce_slots(S_INIT, 3)
        be.pcg_tol_slot(rtol, atol, S_INIT + 1, S_INIT + 2, S_TOL2)
        rz_old, k = S_INIT, 0
        while True:
            done, iters, status = be.flags()          # the only host synchronisation of the loop
            if done or k >= maxit:
                break
            for _ in range(min(CHECK_EVERY, maxit - k)):
                out = S_PAIR + 2 * (k & 1)
                self.halo_exchange_raw(mesh, p, cache_view=True)
                be.spmv_dot_slot(op, p, q, p, lo, hi, S_PQ)             # q = A p, local p.q
                self.allreduce_slots(S_PQ, 1)
                be.pcg_xr_slot(xh, r, p, q, dinv, z, lo, hi, rz_old, S_PQ, out)
                self.allreduce_slots(out, 2)                             # (r.z, r.r)
                be.pcg_check_slot(out + 1, S_TOL2)
                be.pcg_p_slot(p, z, lo, hi, out, rz_old)
                rz_old = out
                k += 1
        if status != 0:
            raise RuntimeError("sharded PCG breakdown (NaN residual) after %d iterations" % iters)
        s = be.slots_get(0, 24)
        bb = s[S_INIT + 2]
        rr = s[S_FINAL_RR] if iters > 0 else s[S_INIT + 1]
        x.touched_dev()
        self.halo_exchange(mesh, x)
        return iters, (np.sqrt(rr / bb) if bb > 0 else 0.0)


def slab_ranges(n_planes, world):
    """Owned vertex planes [z0, z1) of every rank: as even as possible, every rank >= 1 plane."""
    if world > n_planes:
        raise ValueError("more ranks (%d) than vertex planes (%d)" % (world, n_planes))
    base, rem = divmod(n_planes, world)
    out, z = [], 0
    for r in range(world):
        k = base + (1 if r < rem else 0)
        out.append((z, z + k))
        z += k
    return out


def sharded_box_mesh(comm, p0, p1, nx, ny, nz):
    """This rank's slab of dolfin.BoxMesh(p0, p1, nx, ny, nz) as a frontend Mesh with a Partition."""
    nx, ny, nz = int(nx), int(ny), int(nz)
    z0, z1 = slab_ranges(nz + 1, comm.world)[comm.rank]
    zf = z0 - 1 if comm.rank > 0 else z0                       # first local plane (ghost below)
    zl = z1 if comm.rank < comm.world - 1 else z1 - 1          # last local plane (ghost above)
    coords, cells = fem.box_mesh_arrays(p0, p1, nx, ny, nz, zf, zl)
    plane = (nx + 1) * (ny + 1)
    lo_g = plane if comm.rank > 0 else 0
    hi_g = plane if comm.rank < comm.world - 1 else 0
    own0 = lo_g
    own1 = own0 + (z1 - z0) * plane
    part = fem.Partition(comm, own0, own1, plane * (nz + 1), lo_g, hi_g, zf * plane)
    part.plane = plane                                         # vertices per z-plane (the slab V-cycle needs the lattice)
    if os.environ.get("PGD_HALO_DIRECT", "0") in ("1", "probe") and hasattr(comm, "enable_direct_halo") and getattr(comm.be, "name", "") == "hip":
        comm.enable_direct_halo(own1 + hi_g, own0, own1, lo_g, hi_g, use=os.environ["PGD_HALO_DIRECT"] == "1")
    mesh = fem.Mesh(coords, cells, part)
    mesh._global_box = (p0, p1, nx, ny, nz)                    # what unshard() rebuilds the whole mesh from
    mesh._on_boundary = fem.box_hull_mask(nx, ny, nz, zf, zl)
    lo_c = np.array([p0.x(), p0.y(), p0.z()]) if hasattr(p0, "x") else np.asarray(p0, dtype=float)
    hi_c = np.array([p1.x(), p1.y(), p1.z()]) if hasattr(p1, "x") else np.asarray(p1, dtype=float)
    tol = 1e-9 * float(np.max(np.abs(hi_c - lo_c)))
    mesh._hull_test = lambda X: np.any((np.abs(X - lo_c) <= tol) | (np.abs(X - hi_c) <= tol), axis=1)      # points on the hull of the WHOLE box
    assert mesh.num_vertices() == own1 + hi_g
    return mesh


def gather_owned(comm, mesh, local_values):
    """All ranks' owned entries concatenated in global vertex order (tests / output only)."""
    torch, dist = comm.torch, comm.dist
    part = mesh.part
    mine = np.ascontiguousarray(local_values[part.own0:part.own1])
    sizes = [None] * comm.world
    dist.all_gather_object(sizes, int(mine.size))
    out = [None] * comm.world
    dist.all_gather_object(out, mine)
    return np.concatenate(out)


def unshard(problem, root=0):
    """After solve_PGD with a row-sharded spatial mesh: on rank `root` the sharded dimension of `problem` is replaced by the WHOLE
    mesh with the modes gathered from the ranks, so that everything downstream of the solve - return_PGD(), PGD.evaluate, the error
    computation, the result files (pgdrome_amd/model.py, io.py: /root/reference/pgdrome/model.py:162-575, 724-953) - works as after a
    one-process run.  Output only: the gathered modes live on the host / on root's GPU (0.13 GB per mode at 256^3).  P1 spaces, scalar
    or vector-valued (a slab numbers its P2 nodes plane by plane: not gathered).  Collective; returns True on `root`, False elsewhere
    (where the problem is left as it was)."""
    done = False
    for d, V in enumerate(problem.V):
        mesh = V.mesh()
        part = getattr(mesh, "part", None)
        if part is None:
            continue
        if V.ufl_element().degree() != 1:
            raise NotImplementedError("unshard: P2 on a sharded mesh is not gathered")
        comm, nc = part.comm, int(getattr(V, "_ncomp", 1))
        gathered = []
        for f in problem.PGD_func[d]:
            mine = np.ascontiguousarray(np.asarray(f.vector()[:])[nc * part.own0:nc * part.own1])
            box = [None] * comm.world if comm.rank == root else None
            comm.dist.gather_object(mine, box, dst=root)
            gathered.append(np.concatenate(box) if comm.rank == root else None)
        if comm.rank != root:
            continue
        p0, p1, nx, ny, nz = mesh._global_box
        whole = fem.BoxMesh(p0, p1, nx, ny, nz)
        gV = fem.VectorFunctionSpace(whole, "CG", 1, dim=nc) if nc > 1 else fem.FunctionSpace(whole, "CG", 1)
        funcs = []
        for old, values in zip(problem.PGD_func[d], gathered):
            g = fem.Function(gV)
            g.vector()[:] = values
            if hasattr(old, "name") and callable(getattr(old, "rename", None)):
                try:
                    g.rename(old.name(), old.name())
                except Exception:       # noqa: BLE001 - a label only
                    pass
            funcs.append(g)
        problem.PGD_func[d] = funcs
        problem.V[d] = gV
        problem.meshes[d] = whole
        done = True
    return done
