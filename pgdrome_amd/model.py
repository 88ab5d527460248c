"""Result container returned by ``PGDProblem.return_PGD()``.

Counterpart of the reference's ``pgdrome.model.PGD`` as far as the hot path
touches it (/root/reference/pgdrome/solver.py:883-907 builds it; model.py:25-160
stores meshes and modes; model.py:724-860 ``evaluate`` reconstructs
u(fixed dim) = sum_k F_fixed^k prod_i F_i^k(x_i)).  The post-processing surface
(pxdmf/hdf5 I/O, sensor responses, error computation) is SURVEY section 8(f)
"next" work and not built yet.
"""
from __future__ import annotations

import numpy as np

from . import fem


class PGDMesh:
    def __init__(self, name, fmesh):
        self.name = name
        self.fmesh = fmesh
        self.dim = fmesh.topology().dim()
        self.numNodes = fmesh.num_vertices()
        self.numElements = fmesh.num_cells()
        X = fmesh.coordinates()
        self.dataX = X[:, 0].copy()
        self.dataY = X[:, 1].copy() if X.shape[1] > 1 else np.zeros(self.numNodes)
        self.dataZ = X[:, 2].copy() if X.shape[1] > 2 else np.zeros(self.numNodes)
        self.topology = fmesh.cells()
        self.attributes = []


class PGDAttribute:
    def __init__(self, name, n_modes, field="Node", ftype="Scalar"):
        self.name, self._type, self.field, self.n_modes = name, ftype, field, n_modes
        self.interpolationfct = []   # the mode Functions
        self.data = []               # vertex values per mode


class PGD:
    def __init__(self, name=None, n_modes=0, fmeshes=[], pgd_modes=[], name_coord=[], modes_info=[],
                 verbose=False, *args, **kwargs):
        self.name = name
        self.numModes = n_modes
        self.used_numModes = n_modes
        self.problem = None
        self.mesh = []
        for d, fm in enumerate(fmeshes):
            pm = PGDMesh("PGD%d" % (d + 1), fm)
            pm.info = [name_coord[d]] if d < len(name_coord) else []
            att = PGDAttribute(modes_info[0] if modes_info else "u", n_modes,
                               modes_info[1] if len(modes_info) > 1 else "Node",
                               modes_info[2] if len(modes_info) > 2 else "Scalar")
            att.interpolationfct = list(pgd_modes[d])
            att.data = [f.compute_vertex_values() for f in pgd_modes[d]]
            pm.attributes.append(att)
            self.mesh.append(pm)

    def print_info(self):
        print("PGD solution %r: %d modes, meshes %s" % (self.name, self.numModes, [m.numNodes for m in self.mesh]))

    def evaluate(self, fixed_dim, free_dim, coord, attri):
        """Function on mesh `fixed_dim` for the given coordinates of the other dimensions."""
        if len(free_dim) != len(coord):
            raise ValueError("number of free dimensions and coordinates differ")
        V = self.mesh[fixed_dim].attributes[attri].interpolationfct[0].function_space()
        out = fem.Function(V)
        acc = np.zeros(V.dim())
        for k in range(self.used_numModes):
            fac = 1.0
            for i, d in enumerate(free_dim):
                fac *= self.mesh[d].attributes[attri].interpolationfct[k](coord[i])
            acc += fac * self.mesh[fixed_dim].attributes[attri].interpolationfct[k].compute_vertex_values()
        out.vector()._host = acc
        out.vector().touched_host()
        return out
