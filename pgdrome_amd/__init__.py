"""pgdrome_amd - MI355X-native PGD enrichment engine.

``pgdrome_amd.solver.PGDProblem`` is the drop-in counterpart of
``pgdrome.solver.PGDProblem`` (BAMresearch/PGDrome); ``pgdrome_amd.fem`` is the
subset of the dolfin API its weak-form callbacks use; all numerics run in
hand-written HIP kernels for gfx950 behind the C-ABI of include/pgd_amd.h.
"""
__version__ = "0.1.0"
