/*
 * pgd_amd.h - C ABI of the MI355X-native PGD fixed-point engine (libpgd_amd.so).
 *
 * This is the drop-in boundary for the hot path of BAMresearch/PGDrome:
 * PGDProblem.solve_PGD -> FP_solve -> one FEM assemble + solve per separated
 * dimension (reference pgdrome/solver.py:306-506, 508-881).  The reference has
 * no FFI of its own - it is pure Python that delegates every numeric step to
 * FEniCS (dolfin/PETSc/MUMPS).  Each entry point below therefore names the
 * reference call site(s) whose delegated native stage it replaces; the ctypes
 * binding a maintainer would add on the reference side is in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; all objects are opaque 64-bit handles (> 0 valid);
 *   - every function returns 0 (PGD_OK) or a negative error code, never throws
 *     or aborts across the ABI; pgd_last_error() gives the message;
 *   - host buffers are caller-owned, C-contiguous (f64 / i32) and are copied;
 *     device buffers are library-owned and freed by the matching *_free;
 *   - one context per device and process; calls on one context are serialised
 *     by the caller; all work is enqueued on the context's single HIP stream
 *     (the caller may hand in its own stream, e.g. torch's current stream, so
 *     RCCL collectives issued through torch.distributed order with it);
 *   - all arithmetic is float64, indices are int32 (nnz < 2^31 is checked).
 */
#ifndef PGD_AMD_H
#define PGD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t pgd_handle;

enum {
    PGD_OK = 0,
    PGD_ERR_INVALID = -1,   /* bad handle / argument / shape mismatch            */
    PGD_ERR_HIP = -2,       /* a HIP runtime call failed                         */
    PGD_ERR_NOMEM = -3,
    PGD_ERR_LIMIT = -4,     /* a structural limit was exceeded (row too long...) */
    PGD_ERR_SINGULAR = -5,  /* zero pivot / PCG breakdown                        */
    PGD_ERR_NODEVICE = -6,
    PGD_ERR_TIMEOUT = -7,   /* sharded solve: no progress on the stream within the deadline (pgd_comm_timeout)  */
    PGD_ERR_PEER = -8       /* sharded solve: another rank reported a rank-local failure; every rank leaves with an error */
};

/* element-matrix kinds ("atoms"); row = test, col = trial */
enum {
    PGD_ATOM_MASS = 0,    /* int u v                                              */
    PGD_ATOM_STIFF = 1,   /* int grad u . grad v                                  */
    PGD_ATOM_DUDV = 2,    /* int u_{,da} v_{,db}                                  */
    PGD_ATOM_CONV = 3,    /* int u_{,da} v        (time derivative term)          */
    PGD_ATOM_CONVT = 4,   /* int u v_{,db}                                        */
    PGD_ATOM_WMASS = 5,   /* int w u v,  w a P1 vertex field                      */
    PGD_ATOM_WSTIFF = 6   /* int w grad u . grad v                                */
};

/* ----------------------------------------------------------------- context --- */
/* stream: NULL -> the library creates its own HIP stream; otherwise a
 * hipStream_t owned by the caller.                                              */
int pgd_ctx_create(int device, void *stream, pgd_handle *ctx);
int pgd_ctx_destroy(pgd_handle ctx);
int pgd_sync(pgd_handle ctx);
const char *pgd_last_error(pgd_handle ctx);
int pgd_version(void);
int pgd_device_count(void);

/* ------------------------------------------------------------------ meshes --- */
/* Replaces dolfin's Mesh + DofMap + sparsity-pattern build behind
 * FunctionSpace(mesh,"CG",1) (callers: tests/integration/test_heat1D.py:26-40).
 * coords: nv x gdim row-major; cells: nc x nvpc.  P1: nvpc = gdim+1 simplex vertices.  P2
 * (FunctionSpace(mesh,"CG",2), tests/integration/test_elastic.py:33, test_laplace.py:880):
 * nvpc = 3 / 6 / 10 for gdim = 1 / 2 / 3; "coords" are the NODES (vertices and edge midpoints),
 * a cell record = its vertices followed by its edge nodes in the UFC local edge order
 * (interval: (v0, v1, mid); triangle edges (1,2),(0,2),(0,1); tetrahedron (2,3),(1,3),(1,2),
 * (0,3),(0,2),(0,1)); straight-sided geometry is taken from the vertices.
 * Builds on the device: vertex->cell adjacency (sorted) and the CSR pattern of
 * "vertices sharing a cell" with sorted columns.                                */
int pgd_mesh_upload(pgd_handle ctx, const double *coords, int64_t nv, int gdim,
                    const int32_t *cells, int64_t nc, int nvpc, pgd_handle *mesh);
/* Vector-valued Lagrange space on a scalar layout (VectorFunctionSpace(mesh,"P",k),
 * tests/integration/test_solver_problem.py:74): dof (node i, component c) = ncomp*i + c; the CSR
 * pattern couples all components of neighbouring nodes.  The result is a layout like any other
 * (vectors of ncomp*nv entries; pgd_op_combine, pgd_spmv, pgd_pcg_solve, ... apply unchanged); its atoms
 * are built from atoms of the scalar layout with pgd_atom_embed.                                 */
int pgd_mesh_blocked(pgd_handle ctx, pgd_handle scalar_mesh, int ncomp, pgd_handle *mesh);
int pgd_mesh_info(pgd_handle ctx, pgd_handle mesh, int64_t *nv, int64_t *nc, int64_t *nnz,
                  int32_t *max_row_len, int32_t *kl, int32_t *ku);
int pgd_mesh_pattern_download(pgd_handle ctx, pgd_handle mesh, int32_t *row_ptr, int32_t *cols);
/* number of distinct relative column patterns held in the mesh's column dictionary
 * (0: the pattern is too irregular, SpMV streams the column ids)                   */
int pgd_mesh_dict_count(pgd_handle ctx, pgd_handle mesh, int32_t *count);
/* Which form of the SPD product the mesh's patterns allow: slots = 0 (CSR kernels only), 4 or 8 upper slots
 * per row of the symmetric half storage (k_spmv_sym); nx, ny > 0 when the rows also form a full structured
 * vertex grid (row = x + nx y + nx ny z): k_spmv_dia_march2 / k_spmv_diac_march2 march along z with x planes in LDS. */
int pgd_mesh_sym_info(pgd_handle ctx, pgd_handle mesh, int32_t *slots, int32_t *nx, int32_t *ny);
/* *is_lattice = 1 when the mesh is a 3-D structured vertex grid whose coordinates are origin + index * steps[axis] to within
 * 8 ulp (see PGD_TUNE_ASM_LATTICE); steps: 3 doubles (zeros otherwise).                                       */
int pgd_mesh_lattice(pgd_handle ctx, pgd_handle mesh, int32_t *is_lattice, double *steps);
int pgd_mesh_free(pgd_handle ctx, pgd_handle mesh);

/* ----------------------------------------------------------------- vectors --- */
int pgd_vec_alloc(pgd_handle ctx, int64_t n, pgd_handle *vec);
int pgd_vec_free(pgd_handle ctx, pgd_handle vec);
int pgd_vec_size(pgd_handle ctx, pgd_handle vec, int64_t *n);
int pgd_vec_upload(pgd_handle ctx, pgd_handle vec, const double *host, int64_t offset, int64_t count);
int pgd_vec_download(pgd_handle ctx, pgd_handle vec, double *host, int64_t offset, int64_t count);
int pgd_vec_ptr(pgd_handle ctx, pgd_handle vec, void **device_ptr);
int pgd_vec_fill(pgd_handle ctx, pgd_handle vec, double value);
int pgd_vec_copy(pgd_handle ctx, pgd_handle dst, pgd_handle src);
int pgd_vec_scale(pgd_handle ctx, pgd_handle vec, double a);                 /* v *= a      */
int pgd_vec_mul(pgd_handle ctx, pgd_handle y, pgd_handle a, pgd_handle x);   /* y = a .* x entry by entry (y may be x or a): the Jacobi
                                                                               * scaling of the sharded BiCGStab, pgdrome_amd/dist.py */
int pgd_vec_axpy(pgd_handle ctx, pgd_handle y, double a, pgd_handle x);      /* y += a x    */
int pgd_vec_set(pgd_handle ctx, pgd_handle vec, const int32_t *idx, const double *val, int64_t n);
/* y = sum_k coefs[k] * xs[k]: the online reconstruction u = sum_k c_k F^k of a PGD solution
 * on its large dimension (replaces the numpy loop of model.py:822-842).            */
int pgd_vec_lincomb(pgd_handle ctx, pgd_handle y, const pgd_handle *xs, const double *coefs, int k);
/* dot over [lo,hi) (hi < 0 -> whole vector); deterministic two-stage reduction.
 * Replaces Vector.inner / the Euclidean residual norm of solver.py:388.         */
int pgd_vec_dot(pgd_handle ctx, pgd_handle x, pgd_handle y, int64_t lo, int64_t hi, double *out);

/* ------------------------------------------------------------------- atoms --- */
/* Replaces FFC tabulate_tensor + dolfin Assembler for one bilinear "atom" on
 * one separated dimension (triggered from solver.py:636,716 and from every
 * dolfin.assemble inside the callbacks, e.g. test_heat1D.py:59-62).  Values
 * are laid out over the mesh's CSR pattern.  wvec: vertex weights for the
 * weighted kinds, 0 otherwise.  Deterministic (owner-computes, no atomics).     */
int pgd_atom_assemble(pgd_handle ctx, pgd_handle mesh, int kind, int da, int db,
                      pgd_handle wvec, pgd_handle *atom);
/* dst[(ncomp*i + cv), (ncomp*j + cu)] += coef * src[i, j]: the scalar atom `src` (e.g. DUDV(a,b)) placed
 * in the (test component cv, trial component cu) block of an atom of the blocked layout.  dst = 0
 * creates a zero atom first; *out is the destination.  One term of inner(C*eps(u), eps(v))*dx
 * (test_solver_problem.py:127-150) is one call.                                                  */
int pgd_atom_embed(pgd_handle ctx, pgd_handle blocked_mesh, pgd_handle src, int cv, int cu, double coef,
                   pgd_handle dst, pgd_handle *out);
int pgd_atom_upload(pgd_handle ctx, pgd_handle mesh, const double *vals, pgd_handle *atom);
int pgd_atom_download(pgd_handle ctx, pgd_handle atom, double *vals);
int pgd_atom_free(pgd_handle ctx, pgd_handle atom);

/* --------------------------------------------------------------- operators --- */
/* A = sum_t coefs[t] * atoms[t] with symmetric Dirichlet elimination of
 * bc_dofs (rows and columns -> identity).  Replaces the per-solve global
 * re-assembly + DirichletBC.apply inside LinearVariationalSolver.solve()
 * (solver.py:627-636, 704-716).  *op == 0 allocates, otherwise the storage of
 * the existing operator (same mesh) is reused.  The result is an atom handle.   */
int pgd_op_combine(pgd_handle ctx, pgd_handle mesh, const pgd_handle *atoms, const double *coefs,
                   int n, const int32_t *bc_dofs, int64_t nbc, pgd_handle *op);

/* y[r0:r1) = (A x)[r0:r1)  (r1 < 0 -> all rows).  The gated kernel k_spmv_csr.  */
int pgd_spmv(pgd_handle ctx, pgd_handle A, pgd_handle x, pgd_handle y, int64_t r0, int64_t r1);
/* out = sum_{i in [r0,r1)} x_i (A y)_i.  Replaces assemble(F*A*G*dx) scalars
 * (callbacks; solver.py:443, 839-841) and dolfin.norm (solver.py:342,754,837).  */
int pgd_bilinear(pgd_handle ctx, pgd_handle A, pgd_handle x, pgd_handle y, int64_t r0, int64_t r1,
                 double *out);
/* out[m] = sum_i x_i (A y_m)_i for m < ny: one pass over A for a whole family of
 * stored modes (the O(n_enr) scalar functionals of rhs_fct, test_heat1D.py:141-165). */
int pgd_bilinear_many(pgd_handle ctx, pgd_handle A, pgd_handle x, const pgd_handle *ys, int ny,
                      int64_t r0, int64_t r1, double *out);

/* ----------------------------------------------------------------- solvers --- */
/* Jacobi-preconditioned CG, x holds the start vector and receives the solution.
 * Stops when ||r||_2 <= max(rtol ||b||_2, atol) on the TRUE residual (PETSc's CG tests the preconditioned
 * norm by default; the reference solves directly, so either is a tolerance on an exact answer).  Replaces
 * PETSc/MUMPS behind solver.solve() (solver.py:592-595, 633-636) for the SPD spatial systems.
 * Reaching maxit is NOT an error here (iters == maxit, relres > rtol: the caller decides - pgdrome_amd.fem raises
 * like dolfin's error_on_nonconvergence).  On an error return (a failing launch or copy in mid-loop) x is handed
 * back in its own coordinates, never in the scaled ones the recurrence works in, and the operator's symmetric
 * copy is dropped.                                                                                         */
int pgd_pcg_solve(pgd_handle ctx, pgd_handle op, pgd_handle b, pgd_handle x, double rtol,
                  double atol, int maxit, int *iters, double *relres);
/* Banded LU with partial pivoting in one workgroup, for the small and possibly
 * non-symmetric 1-D systems (time: u'v) and the FD-mode solve (solver.py:939).  */
int pgd_band_solve(pgd_handle ctx, pgd_handle op, pgd_handle b, pgd_handle x);

/* ------------------------------------------------- distributed PCG pieces --- */
/* Row-sharded solve: the host drives the recurrence and places the RCCL halo
 * exchange / all-reduce between these calls (pgdrome_amd/dist.py).  Scalars
 * live in a device bank of PGD_NSLOTS doubles so nothing round-trips to the
 * host inside an iteration; every kernel is a no-op once the done flag is set. */
#define PGD_NSLOTS 64
int pgd_slots_ptr(pgd_handle ctx, void **device_ptr);
int pgd_slots_download(pgd_handle ctx, double *out, int first, int count);
int pgd_slots_upload(pgd_handle ctx, const double *in, int first, int count);
int pgd_flags_reset(pgd_handle ctx);
int pgd_flags_download(pgd_handle ctx, int32_t *done, int32_t *iters, int32_t *status);
int pgd_op_diag_inv(pgd_handle ctx, pgd_handle op, pgd_handle dinv);
/* Build the symmetric half storage of an SPD operator for the products of a host-driven solve
 * (the library solves do this themselves): *used = 1 if the mesh's patterns qualify and
 * a_ij == a_ji held to rounding, else 0 and the CSR kernels stay in use.          */
int pgd_op_symmetrize(pgd_handle ctx, pgd_handle op, int *used);
/* Which storage form a product with this ATOM reads right now: 0 = the CSR kernels, 1 = the z-march over its diagonal form,
 * 2 = the z-march over its row-class dictionary (one code byte per row).  Forms 1 / 2 exist on structured vertex grids once
 * pgd_op_combine has put the atom into an operator; the first call looks for the atom's row classes (once: atoms do not
 * change).  The frontend asks before it trades a fused product-dot (fem._bilinear_scalar, the functionals of
 * solver.py:547-569) for product + dot with the product kept.  y is bit-identical in all three forms. */
int pgd_atom_product_form(pgd_handle ctx, pgd_handle A, int *form);

/* Look for the lossless row-class dictionary of the operator's diagonal form (structured vertex grids, after
 * pgd_op_symmetrize / pgd_op_combine; PGD_TUNE_SPMV_ROW_CLASSES): *classes = number of distinct 8-tuples of slot values
 * (1..255) when every row was verified bit by bit against its class - the z-march of later products then reads one code
 * byte per row (k_spmv_diac_march2) - or 0: none (more than 255 classes, not a grid, too few planes).  The library solves
 * call this themselves for the scaled operator; any later change of the operator's values drops the dictionary.   */
int pgd_op_classify(pgd_handle ctx, pgd_handle op, int *classes);
/* y = A x on [r0,r1), slot <- sum w_i y_i (local part)                          */
int pgd_spmv_dot_slot(pgd_handle ctx, pgd_handle A, pgd_handle x, pgd_handle y, pgd_handle w,
                      int64_t r0, int64_t r1, int slot);
/* r = b - q, z = dinv r, p = z on [r0,r1); slots s..s+2 <- (r.z, r.r, b.b)       */
int pgd_pcg_init_slot(pgd_handle ctx, pgd_handle b, pgd_handle q, pgd_handle dinv, pgd_handle r,
                      pgd_handle z, pgd_handle p, int64_t r0, int64_t r1, int slot);
/* tol2 slot <- max(rtol^2 S[bb], atol^2); done <- S[rr] <= tol2                  */
int pgd_pcg_tol_slot(pgd_handle ctx, double rtol, double atol, int slot_rr, int slot_bb,
                     int slot_tol2);
/* alpha = S[rz]/S[pq]; x += alpha p; r -= alpha q; z = dinv r;
 * slots out..out+1 <- (r.z, r.r)                                                */
int pgd_pcg_xr_slot(pgd_handle ctx, pgd_handle x, pgd_handle r, pgd_handle p, pgd_handle q,
                    pgd_handle dinv, pgd_handle z, int64_t r0, int64_t r1, int slot_rz,
                    int slot_pq, int slot_out);
/* iters += 1; done <- S[rr] <= S[tol2] (or breakdown)                            */
int pgd_pcg_check_slot(pgd_handle ctx, int slot_rr, int slot_tol2);
/* beta = S[num]/S[den]; p = z + beta p                                           */
int pgd_pcg_p_slot(pgd_handle ctx, pgd_handle p, pgd_handle z, int64_t r0, int64_t r1,
                   int slot_num, int slot_den);

/* Single-reduction (Chronopoulos-Gear) form of the same recurrence: one all-reduce of
 * S[base..base+4] = (r.u, r.r, w.u interior / low / high boundary rows) per iteration.
 * S[base+5] alpha, S[base+6] beta, S[base+7] previous r.u, S[base+8] b.b.              */
int pgd_cg_init_slot(pgd_handle ctx, pgd_handle b, pgd_handle q, pgd_handle dinv, pgd_handle r,
                     pgd_handle u, pgd_handle p, pgd_handle s, int64_t r0, int64_t r1, int base);
/* p = u + beta p; s = w + beta s; x += alpha p; r -= alpha s; u = dinv r; S[base..+1] <- (r.u, r.r) */
int pgd_cg_update_slot(pgd_handle ctx, pgd_handle x, pgd_handle r, pgd_handle u, pgd_handle w,
                       pgd_handle p, pgd_handle s, pgd_handle dinv, int64_t r0, int64_t r1, int base);
/* after the all-reduce: next alpha/beta, iteration count, done <- r.r <= tol2 (init: also sets tol2) */
int pgd_cg_scalars_slot(pgd_handle ctx, int base, int init, double rtol, double atol);

/* ----------------------------------------------- sharded solve, in-library --- */
/* The same single-reduction recurrence with the ITERATION LOOP AND THE COMMUNICATION inside the
 * library (no host language between two iterations): per iteration one halo exchange of the
 * boundary planes with the z-neighbours (rank-1, rank+1) and one all-reduce of 5 slots, issued on
 * the context's stream.  A context is bound to its communication once:
 *   - RCCL (one process per GPU, xGMI): rank 0 calls pgd_comm_unique_id, the 128 bytes travel to
 *     the other ranks by any means (torch.distributed broadcast), every rank calls
 *     pgd_comm_bind_rccl; the binding is checked on the spot (ring shift + all-reduce of rank ids).
 *     librccl is resolved at run time from the copy already loaded in the process.
 *   - callbacks: the library calls back for the two steps (several ranks sharing one GPU in tests:
 *     host-staged gloo).  A callback returns 0 on success.
 * Local numbering of a rank's vectors: [0, lo_ghost) ghost plane below, [own0, own1) owned rows,
 * [own1, own1 + hi_ghost) ghost plane above; own0 == lo_ghost.                                     */
typedef int (*pgd_halo_fn)(void *user, pgd_handle vec, int64_t own0, int64_t own1, int64_t lo_ghost,
                           int64_t hi_ghost);
typedef int (*pgd_allreduce_fn)(void *user, int first_slot, int count);
int pgd_comm_bind_callbacks(pgd_handle ctx, pgd_halo_fn halo, pgd_allreduce_fn allreduce, void *user,
                            int rank, int world);
int pgd_comm_unique_id(pgd_handle ctx, uint8_t *out128);
int pgd_comm_bind_rccl(pgd_handle ctx, const uint8_t *id128, int rank, int world);
/* Halo exchange concurrent with the product of the rows that read no ghost entry: a second
 * communicator (ncclCommSplit) on its own HIP stream, ordered against the compute stream by two events.
 * mode 1: try to enable - COLLECTIVE, call on every rank after all ranks bound successfully; the ranks must then
 * agree (e.g. MIN all-reduce of *state) and call mode 0 everywhere if any rank reports 0.  mode 0: disable.
 * mode -1: query.  *state: 1 = the sharded solve CAN overlap, 0 = the exchange stays on the compute stream.  A solve uses the
 * second stream only above PGD_TUNE_HALO_OVERLAP_MIN_ROWS rows per rank (default: never); mode -2: *state = did the last solve. */
int pgd_comm_overlap(pgd_handle ctx, int mode, int *state);
int pgd_comm_unbind(pgd_handle ctx);
/* Deadline of the host-side waits inside pgd_pcg_solve_sharded (the look at the flags after every chunk of iterations, the
 * closing agreement): the stream is polled with hipEventQuery, and after `seconds` without completion the call returns
 * PGD_ERR_TIMEOUT with "rank r/w: iteration i, last collective <name> #n" in pgd_last_error (also written to stderr) - a
 * neighbour that died, a collective nobody matches.  The stream is then stuck for good: the caller ends the PROCESS
 * (pgdrome_amd/dist.py exits non-zero).  Default 60 s, or PGD_COMM_TIMEOUT_S from the environment at bind time; <= 0: wait for ever. */
int pgd_comm_timeout(pgd_handle ctx, double seconds);
/* Phase timing of the sharded loop (bench.py's N > 1 line): mode 1 = on + reset, 0 = off, -1 = read only.  One iteration per
 * chunk of 16 is bracketed with HIP events on the compute stream.  out[0..7] = samples, then SECONDS summed over the samples:
 * [1] compute stream waiting for the ghost planes after the interior rows' product (exposed halo time), [2] interior product,
 * [3] boundary rows' product, [4] local sums, [5] the all-reduce (incl. waiting for the slowest rank), [6] vector update,
 * [7] host time spent waiting at the chunk boundaries (all chunks, not sampled).  out may be NULL.        */
int pgd_comm_prof(pgd_handle ctx, int mode, double *out8);
int pgd_comm_info(pgd_handle ctx, int *kind /* 0 none, 1 callbacks, 2 rccl */, int *rank, int *world);
int pgd_comm_halo(pgd_handle ctx, pgd_handle vec, int64_t own0, int64_t own1, int64_t lo_ghost,
                  int64_t hi_ghost);
int pgd_comm_allreduce_slots(pgd_handle ctx, int first_slot, int count);

/* DIRECT HALO of the sharded PCG loop (opt-in; no reference counterpart - /root/reference/pgdrome/solver.py:538-540 is serial).  The
 * boundary planes of the search direction are stored straight into the neighbours' ghost planes through mapped device pointers
 * (hipIpcMemHandle between processes; plain addresses inside one) with a sequence number posted behind them, and the product waits
 * for the numbers of its own ghost planes: no RCCL send / receive kernel in the iteration (14 of 54 us on the slab of an 8-GPU rank).
 * pgd_comm_push_export sizes the loop's work vectors for a vector of n rows partitioned [0, lo) ghost | [own0, own1) | ghost and
 * writes PGD_PUSH_BLOB_BYTES describing them; the caller carries every rank's blob to its neighbours (any transport) and hands the
 * lower / upper neighbour's blob (NULL where there is none) to pgd_comm_push_attach - collective over neighbours, ends with a
 * checked exchange; *state = 1 if the direct halo is usable on this rank.  pgd_pcg_solve_sharded uses it when EVERY rank has it
 * for exactly that partition (its setup vote) and the single-sync recurrence runs; otherwise the binding's exchange.  A number that
 * does not arrive within pgd_comm_timeout ends the solve with PGD_ERR_TIMEOUT.  pgd_comm_push: mode 1 / 0 switch it on / off,
 * -1 reads the state, -2 what the last solve did, 2 queues ONE exchange of the loop's search direction as it stands (probes: timing). */
#define PGD_PUSH_BLOB_BYTES 256
int pgd_comm_push_export(pgd_handle ctx, int64_t n, int64_t own0, int64_t own1, int64_t lo_ghost, int64_t hi_ghost,
                         uint8_t *blob /* PGD_PUSH_BLOB_BYTES */);
int pgd_comm_push_attach(pgd_handle ctx, const uint8_t *lower_blob, const uint8_t *upper_blob, int *state);
int pgd_comm_push(pgd_handle ctx, int mode, int *state);
/* DIRECT ALL-REDUCE of the loop's five sums (same opt-in, behind pgd_comm_push_export): all_blobs = every rank's export blob in rank
 * order (world x PGD_PUSH_BLOB_BYTES, world <= 16).  Every rank maps every rank's flag block; in the loop ONE kernel forms the
 * iteration's local sums, stores them into all mailboxes, posts, waits for everybody's and adds the contributions in rank order
 * (the same bits on every rank): neither k_pcg1_sums nor an RCCL kernel is left in the iteration.  Collective, ends with a checked
 * exchange; used by a solve only if every rank voted for it.  pgd_comm_allreduce_direct: mode as pgd_comm_push (2: one direct
 * all-reduce of slots 48 .. 52). */
int pgd_comm_allreduce_attach(pgd_handle ctx, const uint8_t *all_blobs, int *state);
int pgd_comm_allreduce_direct(pgd_handle ctx, int mode, int *state);
/* Jacobi-PCG on the rows [own0, own1) of this rank's slab of A (replaces the KSP solve of
 * solver.py:636,716 for the row-partitioned spatial dimension); b, x are local slab vectors, x holds
 * the start value and returns with current ghost planes.  iters / rel_res are global.            */
int pgd_pcg_solve_sharded(pgd_handle ctx, pgd_handle A, pgd_handle b, pgd_handle x, int64_t own0,
                          int64_t own1, int64_t lo_ghost, int64_t hi_ghost, double rtol, double atol,
                          int maxit, int *iters, double *rel_res);

/* BiCGStab with Jacobi scaling on the CSR product, for operators that are NOT symmetric (a convection atom
 * u.dx(a) * v * dx on a 2-D / 3-D space, or a 1-D system too long for pgd_band_solve): replaces the MUMPS solve of
 * LinearVariationalSolver for such systems (solver.py:627-636, 704-716).  x holds the start value; stop test
 * ||b - A x|| <= max(rtol ||b||, atol), confirmed on the true residual; PGD_ERR_SINGULAR on repeated breakdown.  */
int pgd_bicgstab_solve(pgd_handle ctx, pgd_handle A, pgd_handle b, pgd_handle x, double rtol, double atol,
                       int maxit, int *iters, double *rel_res);

/* Gram data of the Galerkin start of a PCG solve (the warm start x0 = sum_j c_j v_j with G c = g; this library's
 * addition in front of the solve that replaces solver.py:636,716): out[i*k + j] = v_i . (A v_j) over rows
 * [r0, r1), out[k*k + j] = v_j . b, k <= 17.  k products from the operator's fastest storage form, every dot on
 * the device, ONE host synchronisation; a sharded caller all-reduces `out` once.                        */
int pgd_start_gram(pgd_handle ctx, pgd_handle A, const pgd_handle *vecs, int k, pgd_handle b, int64_t r0,
                   int64_t r1, double *out);
/* r = b - sum_j coefs[j] (A v_j), j < k, from the products A v_j the library still holds from the pgd_start_gram call
 * that came right before (same operator, all rows, k <= 9): the residual of the Galerkin start x0 = sum_j coefs[j] v_j
 * without another product - input of the second level of the start over the spectral vectors
 * (pgdrome_amd/spectral.py; the solve it prepares replaces solver.py:636,716).  PGD_ERR_INVALID if the products are
 * not held.                                                                                               */
int pgd_start_residual(pgd_handle ctx, pgd_handle A, int k, const double *coefs, pgd_handle b, pgd_handle r);

/* out[j] = x . y_j over entries [lo, hi) for k <= 256 vectors: ceil(k / 17) passes over x and ONE host
 * synchronisation.  Serves the functionals of one iterate against all stored modes of its dimension
 * (the scalar assemble() calls inside solver.py:568-612 when the products A y_j are already stored). */
int pgd_vec_multidot(pgd_handle ctx, pgd_handle x, const pgd_handle *ys, int k, int64_t lo, int64_t hi,
                     double *out);

/* Two left factors against the same k <= 128 vectors: out[j] = x0 . y_j, out[k + j] = x1 . y_j over [lo, hi); every y_j is
 * read once for both (ceil(k / 16) passes over x0 and x1), ONE host synchronisation.  The functionals of an iterate F against
 * the stored modes m_j of its dimension under two symmetric atoms (solver.py:568-612: F^T K m_j and F^T M m_j inside the
 * callbacks) as (K F) . m_j and (M F) . m_j: k + 2 vector reads where F . (K m_j), F . (M m_j) take 2 k + 1.            */
int pgd_vec_multidot_pair(pgd_handle ctx, pgd_handle x0, pgd_handle x1, const pgd_handle *ys, int k, int64_t lo,
                          int64_t hi, double *out);

/* ------------------------------------------------------------------ tuning --- */
/* Launch-shape knobs; they change speed (and the order of the dot's partial
 * sums), never which result is computed (PGD_TUNE_FAULT_ITERATION excepted: a test hook).  */
enum {
    PGD_TUNE_PUSH_IN_UPDATE = 49, /* direct halo (pgd_comm_push_*): 1 (default) the boundary planes of the new search direction leave from the
                                update kernel of the iteration itself - no launch for the exchange at all; 0: k_halo_push in front of
                                every product.  Same stores, same iterates. */
    PGD_TUNE_STENCIL_ROWS = 48, /* rows per thread of k_spmv_stencil_march: 4 (64 x 16 patches), 2 (64 x 8: twice the patches per plane, marches
                                 * twice as long on thin z-slabs), 0 (default): 2 where four-row patches would march fewer than 8 planes.   */
    PGD_TUNE_DIA_MARCH3 = 47, /* 1: the plain z-march of the diagonal form (variant 0: no row classes - variable coefficients, graded meshes) runs
                               * in k_spmv_dia_march3: buffer addressing with lane offsets that never change, the next plane's slot values
                               * loaded while the current plane is multiplied; bit-identical to k_spmv_dia_march2 (0, the default: half the vector
                               * instructions, no faster - the march runs at 0.86-0.92 of what the memory system gives ANY kernel for its nine
                               * streams, profiles/r04_dia_march_counters.txt).                                                            */
    PGD_TUNE_SHARD_ONE_MARCH = 46, /* 1 (default): where the halo exchange of pgd_pcg_solve_sharded runs in stream order and the rank's operator is one
                                stencil whose ghost planes hold the same eliminated nodes as its own (checked per solve), the product is ONE march
                                over all owned planes with the ghost planes staged as data; 0: interior march + the boundary planes in row
                                order (what the overlapped exchange always does).  The same y; the fused dots are grouped differently. */
    PGD_TUNE_HALO_OVERLAP_MIN_ROWS = 45, /* pgd_pcg_solve_sharded sends the halo exchange of its products through the second communicator and
                                stream (pgd_comm_overlap) only where the ranks own at least this many rows on average (default 2^40: never;
                                environment: PGD_HALO_OVERLAP_MIN_ROWS; a property of the current binding).  Measured with one rank as its own
                                neighbour: the second stream loses 15 - 19 us per iteration against the exchange in stream order + one march
                                over all owned planes on the slabs of 8-, 4- and 2-GPU ranks of the 256^3 grid; it pays only where the wire
                                adds more than that.  Decided from the all-reduced row count: the same on every rank. */
    PGD_TUNE_COMM_SELF_PERIODIC = 44, /* tests only: with ONE rank, pgd_pcg_solve_sharded and pgd_comm_halo accept ghost planes on both sides
                                and the rank is its own neighbour - the ghost plane below receives the rank's top plane, the one above its
                                bottom plane (a problem periodic in z): the halo communicator, its stream and events, the overlap with the
                                interior rows' product and real RCCL send / receive run inside the iteration loop on a single GPU */
    PGD_TUNE_PCG_DERIVE_SCALED = 43, /* 1 (default): where the operator of pgd_pcg_solve is itself ONE stencil + eliminated nodes on the whole
                                grid (every row verified, PGD_TUNE_SPMV_STENCIL) the couplings of D^-1/2 A D^-1/2 are derived from A's with
                                the arithmetic the scaling pass would apply to every row - no pass over the slot arrays, no second
                                classification, the slot arrays keep A.  0: scale the slot arrays and classify them (the same couplings,
                                bit for bit). */
    PGD_TUNE_MG_MARCH_MIN = 42, /* multigrid levels with at least this many nodes along x and y run their stencil passes in
                                k_spmv_stencil_march (default 64); smaller ones in the plain kernels of pgd_mg.hip */
    PGD_TUNE_MG_CHUNK = 41, /* PCG iterations queued between two looks at the convergence flags when the multigrid preconditioner is on
                                (even, default 2: an iteration is ~50 launches, the next chunk is queued while the flags of the last travel; the Jacobi form queues 16) */
    PGD_TUNE_PCG_PRECOND = 40, /* preconditioner of pgd_pcg_solve: 0 (default) Jacobi = the symmetric diagonal scaling; 1 a geometric
                                multigrid V(1,1) cycle on the scaled operator WHERE it is one stencil on a lattice whose eliminated
                                nodes are exactly the hull (every row verified, pgd_mg.hip), Jacobi everywhere else.  Changes the
                                iterates (same stop test, same tolerance), not the system solved.  The frontend sets it from
                                settings["preconditioner"] (solver.py:593-594: forwarded to the linear solver). */
    PGD_TUNE_CLS_CACHE = 39, /* 1 (default): a mesh remembers the class codes of the operators classified on it (by the signature of their
                                Dirichlet set, scaled or not): the next operator with that structure - the same atoms with other
                                coefficients, every solve of a fixed-point pass - copies the codes, rebuilds the table from the classes'
                                representative rows and verifies EVERY row against its class bit by bit (any mismatch: full classification);
                                0: always the full classification */
    PGD_TUNE_STENCIL_DEPTH = 38, /* plane fetches in flight per workgroup of k_spmv_stencil_march: 3 or 6 (0, default: chosen by the launcher) */
    PGD_TUNE_STENCIL_WG_PER_CU = 37, /* resident workgroups per CU assumed for k_spmv_stencil_march (default 2): sets the march length */
    PGD_TUNE_SPMV_ZCHUNK_STENCIL = 36, /* > 0: planes per march of k_spmv_stencil_march; 0 (default): as many as fill every resident
                                workgroup slot exactly once, in whole groups of the fetch depth */
    PGD_TUNE_SPMV_STENCIL = 35, /* 1 (default): where the row classes of an operator are ONE 8-tuple plus the rows it becomes next to eliminated
                                (Dirichlet) nodes and the rim of the grid - every row and slot verified bit by bit - the z-march takes the
                                couplings from scalar registers, four rows per thread, and reads the code byte only where the codes of a plane
                                differ from the plane below (k_spmv_stencil_march: 16 B per row); bit-identical y; 0: the dictionary form */
    PGD_TUNE_FAULT_STALL_MS = 34, /* tests only: the next pgd_pcg_solve_sharded queues, once, a kernel that spins for this many
                                milliseconds (bounded: at most 20 000) in front of its first chunk - a stream that makes no progress,
                                for the deadline of pgd_comm_timeout */
    PGD_TUNE_FAULT_STAGE = 33, /* tests only: where the NEXT pgd_pcg_solve_sharded fails on this rank, once: 1 = right after the setup
                                vote, 2 = between the setup's halo exchanges, 3 = between the setup all-reduce and the loop, 4 = after the
                                loop, before the closing agreement (0: nowhere; PGD_TUNE_FAULT_ITERATION: inside the loop) */
    PGD_TUNE_SPMV_ZCHUNK_CODED2 = 32, /* k_spmv_diac_march2 on grids with at least 8 planes of work per resident workgroup slot (two per CU): marches long
                                enough that the launch fills every slot exactly once, at most this many planes (default 96, whole sixes; 0: the
                                PGD_TUNE_SPMV_ZCHUNK_CODED rule everywhere) */
    PGD_TUNE_PCG_PIPELINE = 31, /* 1 (default): pgd_pcg_solve and pgd_pcg_solve_sharded queue the next 16-iteration chunk before the host looks at the
                                flags of the previous one (snapshots into pinned memory, one event each): the GPU does not wait for the
                                host's round trip; a chunk queued behind the iteration that converged consists of no-op launches.  Same
                                iterates, same iteration count.  0: look, then queue */
    PGD_TUNE_PCG_EXACT_PHASE = 30, /* 1 (default): the single-sync recurrence of pgd_pcg_solve_sharded measures the true residual norm (one more
                                vector read per row in the update) only near the end, like pgd_pcg_solve: once d_lb r~.r~ comes within 10^4 of
                                the tolerance, d_lb = (all-reduced sum of d_i^-8)^(-1/8) <= d_min, the same number on every rank; the stop test
                                is the true norm either way.  0: the true norm in every iteration */
    PGD_TUNE_PCG_FOLD_FINISH = 29, /* 1 (default): in the single-sync recurrence of pgd_pcg_solve_sharded every workgroup of the vector update
                                forms the stop decision, alpha and beta itself from the five all-reduced sums (workgroup 0 keeps the books):
                                one launch less per iteration, bit-identical iterates; 0: k_pcg1_finish in a launch of its own */
    PGD_TUNE_LAZY_CSR = 28,  /* 1 (default): where pgd_op_combine can form the operator's diagonal form (structured grids, symmetric atoms) it
                                leaves the CSR values to the first reader that asks for them - the solve, its start and its products read
                                the diagonal form only (1.5 ms less per solve at 256^3); 0: both forms at once */
    PGD_TUNE_ATOM_FAST = 27, /* 1 (default): pgd_spmv with an atom whose diagonal form exists takes the z-march, over the atom's own row
                                classes where it has them (looked for once per atom); 0: always the CSR kernels.  Bit-identical y */
    PGD_TUNE_SPMV_ROWS = 1,  /* rows (= threads) per k_spmv_csr workgroup: 64 (default), 128 or 256 */
    PGD_TUNE_SPMV_GRID_MIN_BYTES = 8, /* ... used when a grid plane of values has at least this many bytes (default 0) */
    PGD_TUNE_PCG_FOLD_REDUCE = 12, /* 1 (default): the scaled recurrence sums its reduction partials inside the vector kernels
                                      (3 dependent launches per iteration instead of 5) for systems of up to 2^20 rows */
    PGD_TUNE_PCG_SCALED = 10,   /* 1 (default): pgd_pcg_solve runs CG on D^-1/2 A D^-1/2 (the same iterates as Jacobi-PCG,
                                   two vector passes per iteration fewer) when the symmetric storage applies; 0: unscaled */
    PGD_TUNE_SPMV_ZCHUNK_FORCE = 7, /* > 0: exactly this many planes per march on any grid size (0: adaptive) */
    PGD_TUNE_PCG_SMALL_ROWS = 26, /* structured grids of up to this many rows take the two-launch form as well (default 2^22: +2 % at 128^3; at 256^3 the redundant sums cost more than the launch they save) */
    PGD_TUNE_PCG_SMALL_SINGLE_SYNC = 25, /* 1 (default): systems of up to 2^20 rows - where the launches, not the bytes, set the pace - run the
                                single-sync recurrence in TWO launches per iteration: the product, and an update kernel whose every workgroup sums
                                the partial sums and forms alpha, beta and the stop decision itself (k_pcg1_step); 0: the two-reduction
                                recurrence in three launches (k_pcg_xr_s2 / k_pcg_p_s2) */
    PGD_TUNE_SPMV_FETCH_DEPTH = 24, /* k_spmv_diac_march2: plane fetches in flight per workgroup, 6 (default, marches of 12+ planes in whole sixes) or 3 */
    PGD_TUNE_PCG_STREAM_HINTS = 23, /* 1 (default): in the single-sync recurrence q, r and x - vectors no other kernel of the iteration touches -
                                are read and written with non-temporal hints, p (read by the product next) keeps the default policy and so its
                                place in the Infinity Cache; same arithmetic, bit-identical results */
    PGD_TUNE_PCG_LAG_X = 22,   /* 1 (default): in the single-sync recurrence of pgd_pcg_solve x - an output accumulator that nothing of the
                                recurrence reads - is updated every OTHER iteration with two terms, the earlier direction taken back out of
                                p = r + beta' p' (5 + 7 instead of 7 + 7 vector passes per two iterations); residuals, directions, alpha, beta and
                                iteration counts are bit-identical to 0, x agrees to the rounding of its own updates */
    PGD_TUNE_SPMV_ZCHUNK_CODED = 21, /* k_spmv_diac_march2: most planes per workgroup march (default 24; whole threes, fewer while that keeps
                                ~2 launches of workgroups per slot); PGD_TUNE_SPMV_ZCHUNK_FORCE overrides it too */
    PGD_TUNE_ASM_LATTICE = 20, /* 1 (default): on a 3-D structured vertex grid whose coordinates are origin + index * step per axis (to 8 ulp,
                                checked at pgd_mesh_upload) the P1 assembly takes every edge component as a whole number of steps instead of
                                the difference of two rounded coordinates: congruent cells get identical local matrices, the rows of a uniform
                                grid repeat bit for bit (relative change of the entries ~ 1e-16) - read off the vertex INDICES where every cell spans at most
                                one step per axis (no coordinate is gathered; the same numbers), and where the cells are numbered regularly - cell 6 q + t =
                                tetrahedron t of cube q - the unweighted atoms gather nothing at all (k_assemble_p1_regular); 3: index steps in the general
                                kernel; 2: the steps from the rounded coordinates (r03);
                                0: coordinate differences as they are */
    PGD_TUNE_SPMV_ROW_CLASSES = 19, /* 1 (default): after scaling, pgd_pcg_solve(_sharded) looks for a lossless ROW-CLASS dictionary of the
                                operator's diagonal form (uniform grids repeat a few 8-tuples of slot values; every row verified bit by
                                bit against its class, at most 255 classes, otherwise none) and the z-march then streams one code byte
                                per row instead of 56 B of slot values (k_spmv_diac_march2): same values, same order, same bits */
    PGD_TUNE_PCG_SINGLE_SYNC = 18, /* 1 (default): scaled recurrence on structured grids above 2^20 rows with ONE reduction and ONE
                                      vector kernel per iteration: the product also leaves q.q, beta comes from
                                      r'.r' = alpha^2 q.q - r.r (exact in exact arithmetic; every alpha and the stop test use the
                                      measured r.r); 7 vector passes and 3 launches per iteration instead of 8 and 5.  0: off */
    PGD_TUNE_UNIT_DIAG = 17,   /* 1 (default): the scaled operator D^-1/2 A D^-1/2 of pgd_pcg_solve(_sharded) gets its diagonal set to
                                  exactly 1 on structured grids and the products do not load it (7 instead of 8 slot values per row);
                                  0: diagonal s_i^2 a_ii stored and loaded */
    PGD_TUNE_PCG_DEFER_X = 16, /* 1 (default): in the scaled recurrence on systems above 2^20 rows the update x += alpha p is done by
                                  the kernel that forms p = r + beta p (which reads p anyway): 8 vector passes per iteration instead
                                  of 9, bit-identical iterates; 0: the x / r kernel + p kernel pair */
    PGD_TUNE_FAULT_ITERATION = 15, /* tests only: pgd_pcg_solve_sharded fails on this rank in that iteration, once (-1: never) -
                                      the other ranks must come out of the solve with an error instead of waiting for it */
    PGD_TUNE_COMBINE_DIA = 14, /* 1 (default): on structured vertex grids pgd_op_combine also forms the operator's diagonal
                                  (symmetric half) storage from the atoms' diagonal forms; 0: converted from CSR per solve */
    PGD_TUNE_SPMV_VARIANT = 13, /* the z-march: 0 (default) k_spmv_dia_march2, 64 x 8 patches, 256 threads, two rows per thread;
                                   1: k_spmv_dia_march<8>, 64 x 8 patches, 512 threads; 2: k_spmv_dia_march<4>, 64 x 4 patches, 256 threads */
    PGD_TUNE_SPMV_ZCHUNK = 6,   /* k_spmv_dia_march* (structured vertex grids, x planes in LDS): most planes per
                                   workgroup march (default 8; fewer while that keeps 4 workgroups per CU); 0 = off */
    PGD_TUNE_SPMV_SYM = 3,   /* 1 (default): the products of the SPD solves (pgd_pcg_solve, pgd_pcg_solve_sharded,
                                pgd_spmv_dot_slot after pgd_op_symmetrize) read the operator from its symmetric
                                half storage when the mesh qualifies (k_spmv_sym); 0: always the CSR kernels */
    PGD_TUNE_SPMV_DICT = 2   /* 1 (default): decode column ids from the mesh's relative-pattern
                                dictionary when it has one (k_spmv_csr_dict16 for rows <= 16 entries, else
                                k_spmv_csr_dict); 2: dictionary, generic kernel only; 0: always stream them */
};
int pgd_tune(pgd_handle ctx, int knob, int64_t value);

/* --------------------------------------------------------------- measuring --- */
/* HIP-event timing of k_spmv_csr launches on the context's stream (SURVEY.md
 * section 8d: roofline.achieved = algorithmic bytes / launch time).
 * on = 1: every launch; on = 2: only the PCG instance (fused dot, stores y).     */
int pgd_prof_enable(pgd_handle ctx, int on);
int pgd_prof_read(pgd_handle ctx, int64_t *launches, double *seconds, double *alg_bytes);
/* ... and the least bytes the timed kernels must move in the storage form they actually read (diagonal /
 * symmetric half storage: 8 W + 16..18 B per row; CSR forms: 12 or 8 B per entry + 20..22 B per row): the
 * physical numerator of roofline.frac when the kernel does not stream the CSR arrays.              */
int pgd_prof_read_own(pgd_handle ctx, double *own_bytes);
/* ... and the same for the vector update of the single-sync recurrence (k_pcg1_update: x += alpha p, r -= alpha q,
 * p = r + beta p and the partial sums of r.r in ONE kernel): launches timed, their seconds, and 56 B per row (4 vectors
 * read, 3 written) - the kernel that takes most of a PCG iteration once the product reads a code byte per row.   */
int pgd_prof_read_update(pgd_handle ctx, int64_t *launches, double *seconds, double *bytes);
/* Timed launches that were NOT counted: queued behind the iteration in which their solve converged, every kernel of them returned
 * on the done flag (full bytes, no time - they would bias the averages).  The counts above are the samples that were kept.   */
int pgd_prof_read_dropped(pgd_handle ctx, int64_t *dropped);
/* Seconds a pair of HIP events adds to the kernel it brackets on this context's stream: t(n kernels in one pair) = overhead + n t,
 * measured with n = 1, 2 when the timing is first switched on.  The seconds of pgd_prof_read* are raw event times; bench.py
 * subtracts launches x overhead so that its averages are kernel durations, as rocprofv3 --kernel-trace reports them.       */
int pgd_prof_event_overhead(pgd_handle ctx, double *seconds);
/* Launch counts per product kernel family since the context was created: [0] k_spmv_csr, [1] k_spmv_csr_dict*,
 * [2] k_spmv_sym (row order), [3] k_spmv_dia_rows, [4] k_spmv_dia_march*, [5] k_spmv_multi, [6] k_spmv_diac_march2, [7] k_spmv_stencil_march; tests use them to
 * prove which kernel a call reached, bench.py for its per-kernel breakdown.                          */
int pgd_kernel_counts(pgd_handle ctx, int64_t *out, int n);
/* Row-class classifications since the context was created: done in full (three passes over the slot values with hashing) / served
 * by the mesh's structure cache (codes copied, every row verified against its class: one pass) - PGD_TUNE_CLS_CACHE.   */
int pgd_classify_counts(pgd_handle ctx, int64_t *full, int64_t *cached);
/* Solves that asked for the multigrid preconditioner (PGD_TUNE_PCG_PRECOND = 1) since the context was created: preconditioned by
 * the V-cycle / fallen back to Jacobi because the operator is not one stencil with an eliminated hull.                  */
int pgd_mg_counts(pgd_handle ctx, int64_t *solves, int64_t *fallbacks);
/* The V-cycle of the multigrid preconditioner on a z-slab of a ROW-SHARDED lattice: settings["preconditioner"] = "amg"
 * (forwarded by the reference into its solver, solver.py:593-594, 634-635) on a sharded spatial dimension.  Level 0 stays
 * with the rows (this rank's planes + one ghost plane per side), levels >= 1 are whole on every rank.  The caller owns the
 * PCG loop and the collectives: per cycle  halo(r) -> pgd_mg_slab_down(r, t) -> halo(t) -> pgd_mg_slab_restrict(t, b1) ->
 * all-reduce(b1) -> pgd_mg_coarse(b1, x1) -> pgd_mg_slab_up(r, x1, t, z, slot, &dot) with r . z over the owned rows.
 *   setup: A (unscaled) must be one stencil + eliminated nodes on the owned planes [own0, own1) of the local array (whole planes),
 *   the eliminated nodes exactly the hull of the global lattice of nz_global planes, local plane 0 = global plane z_first;
 *   *applies = 0 where it is not (the caller keeps Jacobi), *n_coarse = entries of a level-1 vector.                      */
int pgd_mg_slab_setup(pgd_handle ctx, pgd_handle A, int nz_global, int z_first, int64_t own0, int64_t own1,
                      int64_t *n_coarse, int *applies);
int pgd_mg_slab_fix_start(pgd_handle ctx, pgd_handle A, pgd_handle b, pgd_handle x, int64_t own0, int64_t own1);   /* x = b on eliminated owned rows */
int pgd_mg_slab_down(pgd_handle ctx, pgd_handle r, pgd_handle t);
int pgd_mg_slab_restrict(pgd_handle ctx, pgd_handle t, pgd_handle b1);
int pgd_mg_coarse(pgd_handle ctx, pgd_handle b1, pgd_handle x1);
int pgd_mg_slab_up(pgd_handle ctx, pgd_handle r, pgd_handle x1, pgd_handle t, pgd_handle z, int slot, double *dot);   /* slot >= 0: r . z into that scalar slot, no host synchronisation (dot may be NULL) */
/* One HIP-event stopwatch on the context's stream (bench.py's micro-sections: N launches between start
 * and stop; stop synchronises on its event).                                                        */
/* Calibration of the PMC byte model: one pass over `vec` with 8- or 16-byte loads (store = 0) or stores (store = 1)
 * per lane - a known byte count to read FETCH_SIZE / WRITE_SIZE against (tools/pmc_calib.py).         */
int pgd_calib_stream(pgd_handle ctx, pgd_handle vec, int bytes_per_lane, int store);
int pgd_timer_start(pgd_handle ctx);
int pgd_timer_stop(pgd_handle ctx, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* PGD_AMD_H */
