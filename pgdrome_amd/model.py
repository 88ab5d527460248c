"""PGD solution container, online evaluation and error computation.

Counterpart of the parts of /root/reference/pgdrome/model.py that sit directly
after the hot path:

* ``PGD`` / ``PGDMesh`` / ``PGDAttribute`` (model.py:25-160, 1573-1663, 1456-1570): what
  ``PGDProblem.return_PGD()`` builds (solver.py:883-907) - per-dimension mesh arrays and the
  stored modes;
* ``PGD.evaluate`` (model.py:724-860): online reconstruction
  ``u(fixed dim) = sum_k F_fixed^k prod_i F_i^k(coord_i)`` - on a large fixed dimension this is a
  tall-skinny product executed on the GPU (``pgd_vec_lincomb``: 8 (K+1) n bytes);
* ``PGD.create_interpolation_fcts`` (model.py:589-722): scipy ``interp1d`` functions of the 1-D
  modes (``interpolationInfo['name'] == 0``) or the mode Functions themselves;
* ``PGDErrorComputation`` (model.py:1666-1825): Latin-hypercube sampling (seed 3452) of the free
  coordinates and relative l2 errors against a full-order model callable.

Not built (out of scope so far): pxdmf / hdf5 / xdmf I/O, sensor responses through
``fenicstools.Probes``, derivatives.
"""
from __future__ import annotations

import logging

import numpy as np
from scipy import interpolate
from scipy.stats import qmc

from . import fem

LOGGER = logging.getLogger(__name__)

DEVICE_EVAL_MIN_DOFS = 1 << 16   # fixed dimensions at least this large are reconstructed on the GPU


class PGDAttribute:
    def __init__(self, name="", n_modes=0, field="Node", _type="Scalar"):
        self.name, self.n_modes, self.field, self._type = name, n_modes, field, _type
        self.data = []                # vertex values per mode, shape (n, 1)
        self.interpolationfct = []    # callables per mode (Functions or interp1d objects)
        self.interpolationInfo = {"name": 1, "family": "P", "degree": 1, "_type": "scalar"}

    def fill_data(self, modes):
        self.interpolationfct = list(modes)
        self.data = [np.asarray(f.compute_vertex_values()).reshape(-1, 1) for f in modes]


class PGDMesh:
    def __init__(self, name, fmesh=None, info=None):
        self.name = name
        self.info = info or []
        self.attributes = []
        self.fenics_mesh = fmesh
        if fmesh is not None:
            X = fmesh.coordinates()
            n = fmesh.num_vertices()
            self.numNodes, self.numElements = n, fmesh.num_cells()
            self.dim = fmesh.topology().dim()
            self.dataX = X[:, 0].copy()
            self.dataY = X[:, 1].copy() if X.shape[1] > 1 else np.zeros(n)
            self.dataZ = X[:, 2].copy() if X.shape[1] > 2 else np.zeros(n)
            self.topology = fmesh.cells()
            self.typElements = {1: "Polyline", 2: "Triangle", 3: "Tetrahedron"}[self.dim]


class PGD:
    def __init__(self, name=None, n_modes=0, fmeshes=[], pgd_modes=[], name_coord=[], modes_info=[],
                 verbose=False, *args, **kwargs):
        self.logger = logging.getLogger(__name__ + "." + self.__class__.__name__)
        self.name = name
        self.numModes = n_modes
        self.used_numModes = n_modes
        self.num_pgd_var = len(fmeshes)
        self.problem = None
        self.mesh = []
        info = list(modes_info) + ["u", "Node", "Scalar"][len(modes_info):]
        for d, fm in enumerate(fmeshes):
            pm = PGDMesh("PGD%d" % (d + 1), fm, [name_coord[d]] if d < len(name_coord) else [])
            att = PGDAttribute(info[0], n_modes, info[1], info[2])
            att.fill_data(pgd_modes[d] if d < len(pgd_modes) else [])
            pm.attributes.append(att)
            self.mesh.append(pm)
        if verbose:
            self.print_info()

    def print_info(self):
        print("PGD solution %r: %d modes, %d coordinates" % (self.name, self.numModes, self.num_pgd_var))
        for m in self.mesh:
            print("  %s %s: %d nodes, %d elements" % (m.name, m.info, m.numNodes, m.numElements))

    # ------------------------------------------------------------------ interpolation
    def create_interpolation_fcts(self, free_dim, attri, verbose=True):
        if len(free_dim) > self.num_pgd_var:
            raise ValueError("given number of Dimensions larger then existing Meshes in PGD solution")
        if attri > len(self.mesh[free_dim[0]].attributes):
            raise ValueError("attribute number not possible")
        for d in free_dim:
            att, pm = self.mesh[d].attributes[attri], self.mesh[d]
            kind = att.interpolationInfo["name"]
            if kind == 0:
                if np.sum(pm.dataY) != 0 and np.sum(pm.dataZ) != 0:
                    raise ValueError("free Dimensions are not 1D, interpolation with INTERP1D not possible")
                how = att.interpolationInfo["kind"]
                att.interpolationfct = [interpolate.interp1d(pm.dataX, att.data[k][:, 0], kind=how)
                                        for k in range(self.numModes)]
            elif kind == 1:
                if len(att.interpolationfct) != self.numModes:
                    raise ValueError("mode functions of dimension %d are missing (loading them from "
                                     "_data.h5 files is not built)" % d)
            else:
                self.logger.error("interpolation name not defined: %s", kind)
        self.logger.info("Attribute interpolation functions saved")

    # --------------------------------------------------------------------- evaluation
    def _check_eval(self, fixed_dim, free_dim, coord, attri):
        if len(free_dim) != self.num_pgd_var - 1:
            raise ValueError("given variables are missing or to much, free_dim=%s <-> num_pgd_var=%s",
                             free_dim, self.num_pgd_var - 1)
        if len(coord) != self.num_pgd_var - 1:
            raise ValueError("given variables are missing or to much, coord=%s <-> num_pgd_var=%s",
                             coord, self.num_pgd_var - 1)
        if len(free_dim) != len(coord):
            raise ValueError("Number of free Dimensions and given coordinates are not the same, "
                             "free_dim=%s <-> coord=%s", free_dim, coord)
        if attri >= len(self.mesh[fixed_dim].attributes):
            raise ValueError("attribute number not possible")
        for d in free_dim:
            if len(self.mesh[d].attributes[attri].interpolationfct) == 0:
                self.create_interpolation_fcts(free_dim, attri)
                break

    def mode_factors(self, free_dim, coord, attri):
        """c_k = prod_i F_i^k(coord_i) for every used mode."""
        c = np.ones(self.used_numModes)
        for k in range(self.used_numModes):
            for i, d in enumerate(free_dim):
                c[k] *= float(self.mesh[d].attributes[attri].interpolationfct[k](coord[i]))
        return c

    def evaluate(self, fixed_dim, free_dim, coord, attri):
        """PGD solution on the fixed dimension for given coordinates of all other dimensions:
        a numpy array (vertex values) in the interp1d mode, otherwise a Function."""
        self._check_eval(fixed_dim, free_dim, coord, attri)
        att = self.mesh[fixed_dim].attributes[attri]
        c = self.mode_factors(free_dim, coord, attri)
        if self.mesh[free_dim[0]].attributes[attri].interpolationInfo["name"] == 0:
            out = np.zeros(att.data[0].shape)
            for k in range(self.used_numModes):
                out += c[k] * att.data[k]
            return out
        modes = att.interpolationfct
        V = modes[0].function_space()
        out = fem.Function(V)
        if V.dim() >= DEVICE_EVAL_MIN_DOFS:
            be = fem.get_backend()
            be.vec_lincomb(out.vector().dev_for_write(), [modes[k].vector().dev() for k in range(self.used_numModes)], c)
            out.vector().touched_dev()
        else:
            acc = np.zeros(V.dim())
            for k in range(self.used_numModes):
                acc += c[k] * modes[k].vector().host()
            out.vector()._host = acc
            out.vector().touched_host()
        return out

    def evaluate_min(self, fixed_dim, free_dim, coord, attri):
        u = self.evaluate(fixed_dim, free_dim, coord, attri)
        return float(np.min(u if isinstance(u, np.ndarray) else u.compute_vertex_values()))

    def evaluate_max(self, fixed_dim, free_dim, coord, attri):
        u = self.evaluate(fixed_dim, free_dim, coord, attri)
        return float(np.max(u if isinstance(u, np.ndarray) else u.compute_vertex_values()))


class PGDErrorComputation(object):
    def __init__(self, fixed_dim=0, n_samples=1, data_test=[], FOM_model=[], PGD_model=[], lim_samples=[],
                 fixed_var=[], *args, **kwargs):
        self.logger = logging.getLogger(__name__ + "." + self.__class__.__name__)
        self.fixed_dim = fixed_dim
        self.n_smp = n_samples
        self.data_test = data_test
        self.FOM_sol = FOM_model
        self.PGD_sol = PGD_model
        self.lim_smp = lim_samples
        self.fixed_var = fixed_var
        self.free_dim = [d for d in range(self.PGD_sol.num_pgd_var) if d not in fixed_dim]

    def sampling_LHS(self):
        """Latin-hypercube samples of the free coordinates, scaled to their ranges (seed 3452)."""
        sample = qmc.LatinHypercube(d=len(self.free_dim), seed=3452).random(n=self.n_smp)
        lo, hi = [], []
        for d in self.free_dim:
            if not self.lim_smp:
                X = self.PGD_sol.problem.meshes[d].coordinates()
                if len(X[0]) == 1:
                    lo.append(float(np.min(X)))
                    hi.append(float(np.max(X)))
                else:
                    print("Not implemented")
            elif len(self.lim_smp[d]) == 2:
                lo.append(float(min(self.lim_smp[d])))
                hi.append(float(max(self.lim_smp[d])))
            else:
                print("Not implemented")
        return qmc.scale(sample, lo, hi).tolist()

    def compute_SampleError(self, u_FOM, u_PGD):
        """|u_PGD - u_FOM|_2 / |u_FOM|_2 for arrays, vertex values or two Functions."""
        if isinstance(u_FOM, np.ndarray):
            pgd = u_PGD.reshape(-1) if isinstance(u_PGD, np.ndarray) else u_PGD.compute_vertex_values()[:]
            return np.linalg.norm(pgd - u_FOM.reshape(-1), 2) / np.linalg.norm(u_FOM.reshape(-1), 2)
        diff = u_FOM.vector().copy()
        diff.axpy(-1.0, u_PGD.vector())
        return diff.norm("l2") / u_FOM.vector().norm("l2")

    def evaluate_error(self):
        if not self.data_test:
            self.data_test = self.sampling_LHS()
        errorL2 = np.zeros(len(self.data_test))
        for i, smp in enumerate(self.data_test):
            if not self.FOM_sol:
                self.logger.error("FEM not defined")
                raise ValueError("FEM not defined")
            u_fem = self.FOM_sol(smp)
            if isinstance(u_fem, float):
                u_fem = np.array(u_fem)
            if not self.PGD_sol:
                self.logger.error("PGD model not defined")
                raise ValueError("PGD model not defined")
            u_pgd = self.PGD_sol.evaluate(int(self.fixed_dim[0]), self.free_dim, smp, 0)
            if not self.fixed_var:
                errorL2[i] = self.compute_SampleError(u_fem, u_pgd)
            else:
                errorL2[i] = self.compute_SampleError(u_fem, np.array([u_pgd(x) for x in self.fixed_var]))
        return errorL2, np.mean(errorL2), np.max(errorL2)
