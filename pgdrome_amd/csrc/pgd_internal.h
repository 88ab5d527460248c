// Internal declarations shared by the translation units of libpgd_amd.so.
// gfx950 (MI355X) only: 64-lane wavefronts, 256 CUs in 8 XCDs, 160 KiB LDS/CU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pgd_amd.h"

namespace pgd {

constexpr int WAVE = 64;
constexpr int TPB = 256;            // threads per workgroup everywhere (4 waves)
constexpr int MAX_VEC_BLOCKS = 2048;  // grid cap of the grid-stride vector kernels (8 per CU)
constexpr int N_XCD = 8;
constexpr int DICT_MAXP = 256;      // patterns a mesh may have for the column dictionary
constexpr int DICT_DLEN = 32;       // max row length representable (ints per pattern, 128 B)
constexpr size_t PAD_BYTES = 256;   // slack after every device array (vector over-reads)

// slots of the device scalar bank used by the library's own PCG loop
enum { S_PQ = 2, S_TOL2 = 5, S_DMIN = 7 /* smallest diagonal entry (scaled PCG) */, S_TMP = 8 /* ..15: batched functionals */ };   // 16..22: PCG r.z / r.r / b.b

struct Ctx;
void dev_release(Ctx *c, void *p, size_t bytes);   // back to the context's buffer pool

struct Obj {
    enum Kind { FREE = 0, VEC, MESH, CSR } kind = FREE;
    Ctx *ctx = nullptr;
    uint64_t serial = 0;        // unique per object of a context (handles are recycled, serials are not): put_obj
    virtual ~Obj() {}
};

struct Vec : Obj {
    double *d = nullptr;
    int64_t n = 0;
    ~Vec() override { if (d) dev_release(ctx, d, (size_t)(n > 0 ? n : 1) * sizeof(double)); }
};

struct Mesh : Obj {
    int gdim = 0, nvpc = 0;
    int ncomp = 1;              // > 1: blocked (vector-valued) layout over the scalar layout `base`: pattern only
    pgd_handle base = 0;
    int64_t nv = 0, nc = 0, nnz = 0;
    double *coords = nullptr;   // SoA: gdim arrays of nv doubles (coalesced per component)
    int4 *cells = nullptr;      // one 16-byte record per cell, unused lanes = -1 (cells of <= 4 nodes)
    int *cellsN = nullptr;      // flat records of nvpc nodes (P2 triangles: 6, P2 tetrahedra: 10)
    int *v2c_ptr = nullptr;     // nv+1
    int *v2c = nullptr;         // nc*nvpc, sorted per vertex
    int *row_ptr = nullptr;     // nv+1
    int *cols = nullptr;        // nnz, sorted per row
    int max_row = 0, kl = 0, ku = 0;
    // Column-index dictionary (lossless): FEM rows repeat a handful of RELATIVE column patterns
    // (cols[k] - row); when a mesh has <= DICT_MAXP of them every row stores a 2-byte pattern id
    // and k_spmv_csr_dict rebuilds its column ids from the table instead of streaming 4 B/entry.
    // Symmetric half storage (pgd_spmv.hip, k_spmv_sym): per relative pattern a 16-int record
    // [packed(ulen, llen, 8 x 3-bit lower slots), 7 upper offsets, 8 lower distances]; sym_w = 4 or 8 upper
    // slots (diagonal first) per row, 0 = the mesh does not qualify (rows longer than 8 + 7, or the slot of a
    // row in its lower neighbours' rows is not a function of its own pattern)
    int *sym_tab = nullptr;
    int sym_w = 0;
    // full structured vertex grid (row = x + nx y + nx ny z, every column a grid neighbour on one of eight diagonals;
    // 0: not one): the slot arrays are then in DIAGONAL form - slot dx + 2 dy + 4 dz of row i = a(i, i + dx + nx dy + nx ny dz)
    // - and the products are k_spmv_dia_march / k_spmv_dia_rows (pgd_spmv.hip)
    int sym_nx = 0, sym_ny = 0;
    // the vertices sit on a uniform lattice origin + index * lat_h (checked at upload, pgd_mesh.hip): the P1 assembly takes
    // edge vectors as whole lattice steps, so congruent cells get identical local matrices
    bool lattice = false;
    bool lattice_unit = false;    // ... and every cell spans at most one step per axis: edge vectors from the vertex indices (k_lattice_cells_verify)
    // ... and the cells are numbered regularly: cell 6 q + t is tetrahedron t of cube q (cubes in vertex order), every cube cut the same way
    // (k_lattice_regular_verify): the cells around a vertex follow from its position - k_assemble_p1_regular gathers nothing
    bool lattice_regular = false;
    int pat_off[6][4] = {{0}};    // vertex offsets of tetrahedron t relative to its cube's first vertex
    int pat_loc[6][8] = {{0}};    // local index of cube corner (dx + 2 dy + 4 dz) in tetrahedron t, -1: not a vertex of it
    double lat_h[3] = {0.0, 0.0, 0.0};
    // What dia_classify learned about operators on this mesh (pgd_spmv.hip): the class CODES of an operator depend on its atoms'
    // structure and its Dirichlet set, not on the coefficients it is combined with - every solve of a fixed-point pass classifies
    // "the same operator with other numbers".  A later classification with the same signature copies the codes, rebuilds the class
    // table from the representative rows and VERIFIES every row against its class bit by bit (+ the class-level stencil relations):
    // one pass over the slot values instead of three with hashing; any mismatch falls back to the full classification.
    struct ClsCache {
        uint64_t sig = 0; int scaled = 0, z_lo = 0, z_hi = 0, ncls = 0;
        uint8_t *codes = nullptr; int *same = nullptr; int *reps = nullptr; size_t bytes = 0;
        bool st_ok = false; int ident = -1, base = -1, zm0 = 0, zm1 = 0;
        uint8_t zero_pat[256];      // per class: bit s = slot s is an exact zero (a coupling to an eliminated node / the rim)
        uint8_t in_range[256];      // per class: it occurs on the verified planes [z_lo, z_hi) (a slab's ghost rows carry classes of their own)
        uint64_t used = 0;
    };
    mutable std::vector<ClsCache> cls_cache;
    mutable uint64_t cls_clock = 0;
    // the Dirichlet list of the last operator combined on this mesh (pgd_op_combine): every solve of a fixed-point pass brings the
    // same list again - compared on the host, word by word, it saves the range check, the upload (1.5 MB at 256^3) and the
    // host synchronisation behind it
    std::vector<int32_t> bc_host;
    int *bc_dev = nullptr;
    size_t bc_dev_bytes = 0;
    uint16_t *pids = nullptr;    // nv
    int *dict_off = nullptr;     // dict_count x DICT_DLEN relative offsets
    int dict_count = 0;          // 0: dictionary not available (irregular pattern) -> plain CSR kernel
    ~Mesh() override {
        for (void *p : {(void *)coords, (void *)cells, (void *)cellsN, (void *)v2c_ptr, (void *)v2c,
                        (void *)row_ptr, (void *)cols, (void *)pids, (void *)dict_off, (void *)sym_tab})
            if (p) (void)hipFree(p);
        for (ClsCache &e : cls_cache) if (e.same) (void)hipFree(e.same);      // (the block starts at `same`)
        if (bc_dev) (void)hipFree(bc_dev);
    }
};

struct Csr : Obj {
    pgd_handle mesh = 0;
    double *vals = nullptr;
    double *dinv = nullptr;    // lazily built inverse diagonal
    bool dinv_valid = false;
    // symmetric half storage of the same operator: sym_w arrays of nv doubles, slot s of row i at
    // uvals[s * nv + i] (slot 0 = diagonal, then the entries right of it; zero padded); on structured vertex grids
    // (Mesh::sym_nx > 0) the slots are the eight diagonals instead
    double *uvals = nullptr;
    bool uvals_valid = false;
    bool uvals_scaled = false;  // the slot arrays hold D^-1/2 A D^-1/2 (inside pgd_pcg_solve only)
    bool uvals_unit = false;    // ... in diagonal form, whose slot 0 is exactly 1 and is not loaded by the products
    int64_t uvals_stride = 0;  // doubles between two slot arrays (rows + padding)
    // row-class dictionary of the CURRENT slot values in diagonal form (dia_classify, pgd_spmv.hip): a code per row and
    // the classes' slot tuples; cls_count = 0: none (every writer of the slot arrays resets it)
    uint8_t *cls = nullptr;
    int *cls_same = nullptr;       // per plane: same codes, row by row, as the plane below
    double *cls_table = nullptr;   // owns the allocation: table, then the codes
    int cls_count = 0;
    uint64_t bc_sig = 0;           // signature of the Dirichlet set the operator was combined with (a hint for Mesh::cls_cache)
    // the classes are ONE stencil + eliminated nodes on the planes [st_z0, st_z1) (dia_classify / k_stencil_verify): couplings
    // c[slot], identity class id; valid exactly as long as cls_count > 0
    bool st_ok = false;
    int st_ident = -1, st_z0 = 0, st_z1 = 0, st_zm0 = 0, st_zm1 = 0;   // (st_zm0 .. st_zm1: the run of planes with identical codes; the rest of the verified planes hold identity rows only)
    double st_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool st_g_lo = false, st_g_hi = false;   // the plane below st_z0 / above st_z1 - 1 is a ghost DATA plane of a sharded slab (k_stencil_ghost)
    bool st_virtual = false;       // st_c holds the couplings of D^-1/2 A D^-1/2 while slots, table and CSR values hold A (pgd_pcg_solve)
    bool cls_tried = false;        // atoms: the dictionary was looked for once for the current slot values (atom_fast_form)
    bool immutable = false;        // an ATOM (assembled / uploaded / embedded): its values change only through pgd_atom_embed(dst)
    uint64_t version = 0;          // bumped by every writer of `vals` after creation
    // pgd_op_combine on a structured grid forms the diagonal form only; the CSR values follow on first use (ensure_vals)
    // from this recipe: atoms by handle + serial + version (a freed or rewritten atom is an error, not a wrong product)
    bool vals_pending = false;
    std::vector<pgd_handle> rec_atoms;
    std::vector<uint64_t> rec_serials, rec_versions;
    std::vector<double> rec_coefs;
    int *rec_bc = nullptr;         // Dirichlet dofs of the combine (device copy)
    int64_t rec_nbc = 0;
    size_t rec_bc_bytes = 0;
    size_t vals_bytes = 0, dinv_bytes = 0, uvals_bytes = 0, cls_bytes = 0;
    ~Csr() override {
        if (rec_bc) dev_release(ctx, rec_bc, rec_bc_bytes);
        if (cls_table) dev_release(ctx, cls_table, cls_bytes);
        if (vals) dev_release(ctx, vals, vals_bytes);
        if (dinv) dev_release(ctx, dinv, dinv_bytes);
        if (uvals) dev_release(ctx, uvals, uvals_bytes);
    }
};

// Communication binding of a context for the row-sharded solve (pgd_comm.hip)
struct Comm {
    int kind = 0;                 // 0 none, 1 host callbacks, 2 RCCL
    int rank = 0, world = 1;
    pgd_halo_fn halo_cb = nullptr;
    pgd_allreduce_fn allreduce_cb = nullptr;
    void *user = nullptr;
    void *nccl = nullptr;         // ncclComm_t
    void *nccl_halo = nullptr;    // second communicator (ncclCommSplit) for the halo exchange on its own stream
    hipStream_t halo_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    bool overlap = false;         // the halo exchange CAN run concurrently with the interior rows' product (second communicator + stream, self-test passed)
    // ... and does, in a solve whose ranks own at least this many rows on average.  Measured (tools/bench_self_periodic.py, one rank as
    // its own neighbour): the second stream costs two event hand-overs per iteration and LOSES 15 - 19 us against the exchange in
    // stream order + one march over all owned planes on the slabs of 8-, 4- and 2-GPU ranks of the 256^3 grid alike (72 / 92 / 130
    // against 54 / 75 / 114 us); what it can hide - a wire latency above that - is unknown here, so the default is "never" and
    // PGD_HALO_OVERLAP_MIN_ROWS / PGD_TUNE_HALO_OVERLAP_MIN_ROWS switch it on.  Decided from the all-reduced row count of the
    // setup vote: the same on every rank
    int64_t overlap_min_rows = (int64_t)1 << 40;
    bool overlap_used = false;    // what the last solve did
    bool self_periodic = false;   // tests only (PGD_TUNE_COMM_SELF_PERIODIC): ONE rank whose ghost planes are its own far boundary planes
    pgd_handle work[7] = {0, 0, 0, 0, 0, 0, 0};   // r, u, w, p, s, q, dinv of the sharded PCG
    int64_t work_n = 0;
    double timeout_s = 60.0;      // deadline of the host-side waits of the sharded solve (pgd_comm_timeout; <= 0: none)
    // boundary snapshots of the sharded loop (pinned): per slot 4 flag ints + the vote slot, and the event behind the copies
    int *snap_flags = nullptr;    // 2 x 4 ints
    double *snap_vote = nullptr;  // 2 doubles (same pinned allocation)
    hipEvent_t snap_ev[2] = {nullptr, nullptr};
    // phase timing (pgd_comm_prof): two sets of 7 markers, alternating with the parity of the chunk
    bool prof = false;
    hipEvent_t mark[2][7] = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
    bool mark_set[2] = {false, false};
    double prof_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // DIRECT halo of the sharded loop's search direction (pgd_comm_push_export / _attach; opt-in): the boundary planes of p are
    // stored straight into the neighbours' ghost planes through mapped pointers (hipIpcMemHandle; the rank itself where it is its
    // own neighbour) and a sequence number posted behind them; the product waits for the numbers of ITS ghost planes.  No RCCL
    // kernel in the iteration.  Geometry of the vector the export was made for: a solve with another one votes the exchange off.
    bool push = false;            // attached and self-tested
    bool push_used = false;       // what the last solve did
    int64_t push_n = 0, push_own0 = 0, push_own1 = 0, push_lo_g = 0, push_hi_g = 0;
    double *push_peer[2] = {nullptr, nullptr};                 // lower / upper neighbour's p (its base address in THIS process)
    int64_t push_peer_n[2] = {0, 0};                           // ... and its length: the upper ghost planes are its last rows
    unsigned long long *push_flags = nullptr;                  // own: [0] posted by the lower neighbour, [1] by the upper one, [2] ticket
    unsigned long long *push_peer_flags[2] = {nullptr, nullptr};
    void *push_mapped[4] = {nullptr, nullptr, nullptr, nullptr};   // what hipIpcOpenMemHandle returned (to close)
    unsigned long long push_seq = 0;
    // DIRECT all-reduce of the loop's five sums (pgd_comm_allreduce_attach; same opt-in, same flag block): every rank stores its sums
    // into every rank's mailbox - block + PUSH_BOX_OFF bytes: [parity][source rank][8 doubles] - posts the sequence number into
    // word PUSH_AR_FLAG0 + its rank of that block, waits for everybody's number in its own, and adds the contributions in rank
    // order: the same bits on every rank.  The iteration's local sums (k_pcg1_sums) are formed by the same kernel.
    bool ar = false, ar_used = false;
    unsigned long long *ar_peer[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *ar_mapped[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long ar_seq = 0;
};
constexpr int PUSH_BLOCK_BYTES = 4096, PUSH_AR_FLAG0 = 8, PUSH_BOX_OFF = 512, PUSH_AR_MAXW = 16;

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::vector<std::unique_ptr<Obj>> objs;   // handle = index + 1
    std::vector<int64_t> free_list;

    // Stream-ordered buffer pool: per-solve operators (2 GB) and work vectors (134 MB) are recycled
    // instead of paying hipMalloc/hipFree (~7 ms each) in every fixed-point pass.  Safe because all
    // work of a context is ordered on its one stream.
    std::multimap<size_t, void *> pool;
    size_t pool_bytes = 0;
    static constexpr size_t POOL_MAX = (size_t)24 << 30;

    double *slots = nullptr;      // PGD_NSLOTS doubles
    int *flags = nullptr;         // [0] done, [1] iters, [2] status
    double *partials = nullptr;   // reduction scratch
    int64_t partials_cap = 0;
    int64_t partials_off = 0;     // the product launchers write their partial sums from here on (several row ranges, one reduction)
    double *work[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t work_cap[7] = {0, 0, 0, 0, 0, 0, 0};
    uint8_t *mask = nullptr;      // Dirichlet column mask scratch
    int64_t mask_cap = 0;
    int *ibuf = nullptr;          // small int32 scratch (bc dofs, index lists)
    int64_t ibuf_cap = 0;
    // pgd_vec_set: the last index list and its values (host copies + what is on the device in ibuf / set_vals): the Dirichlet
    // values of every right-hand side of a fixed-point pass are the same 390 152 (index, value) pairs at 256^3
    std::vector<int32_t> set_idx_host;
    std::vector<double> set_val_host;
    int *set_idx = nullptr;
    double *set_vals = nullptr;
    int64_t set_idx_cap = 0, set_vals_cap = 0;
    int32_t set_idx_max = 0;
    bool set_idx_on_dev = false, set_val_on_dev = false;

    Comm comm;

    int num_cu = 256;
    int spmv_dict = 1;            // use the column dictionary when the mesh has one
    int spmv_rows = 64;           // rows (= threads) per k_spmv_csr workgroup: 64 (default), 128 or 256
    int64_t spmv_grid_min_plane_bytes = 0;   // structured grids whose planes of values are at least this large take the z-march (k_spmv_dia_march*)
    int spmv_zchunk_force = 0;    // > 0: exactly this many planes per march whatever the grid size (tests)
    int fault_iteration = -1;     // tests: pgd_pcg_solve_sharded fails on this rank in that iteration (once)
    int fault_stage = 0;          // tests: ... or at this stage of the solve outside the loop (PGD_TUNE_FAULT_STAGE), once
    int fault_stall_ms = 0;       // tests: ... or its stream stalls for this long in front of the first chunk (PGD_TUNE_FAULT_STALL_MS), once
    int64_t pcg_small_ss_rows = (int64_t)1 << 22;   // ... structured grids up to this many rows as well (PGD_TUNE_PCG_SMALL_ROWS; 128^3: +2 %, 256^3: a loss)
    int pcg_small_ss = 1;         // systems up to 2^20 rows: single-sync recurrence with the scalar step inside the update kernel (2 launches)
    int pcg_stream_hints = 1;     // single-sync recurrence: q, r, x non-temporal, p cached (PGD_TUNE_PCG_STREAM_HINTS)
    int pcg_lag_x = 1;            // single-sync recurrence: x is updated every other iteration, two terms at a time (PGD_TUNE_PCG_LAG_X)
    int asm_lattice = 1;          // lattice meshes: edge vectors as whole lattice steps in the assembly (PGD_TUNE_ASM_LATTICE)
    int spmv_fetch_depth = 6;     // plane fetches in flight per workgroup of k_spmv_diac_march2 (3 or 6; PGD_TUNE_SPMV_FETCH_DEPTH)
    int spmv_zchunk_coded2 = 96;  // ... and where every slot (two workgroups per CU) gets at least 24 planes: marches that fill the slots exactly once, at most this long (0: off)
    int spmv_zchunk_coded = 24;   // most planes per march of k_spmv_diac_march2 (PGD_TUNE_SPMV_ZCHUNK_CODED)
    int pcg_exact_phase = 1;      // sharded single-sync loop: the true r.r only near the end (one vector read per row less in the update)
    int pcg_fold_finish = 1;      // sharded single-sync loop: the scalar step inside the update kernel (one launch less per iteration)
    int atom_fast = 1;            // products with an atom whose diagonal form exists take the z-march (+ its own row classes, looked for once)
    int lazy_csr = 1;             // pgd_op_combine forms only the diagonal form where it can; CSR values on first use
    uint64_t next_serial = 1;
    int pcg_precond = 0;          // pgd_pcg_solve: 0 = Jacobi (the diagonal scaling), 1 = geometric multigrid V-cycle where the scaled operator is one
                                  // stencil on a lattice whose eliminated nodes are exactly its hull (pgd_mg.hip); anything else falls back to 0
    struct Mg *mg = nullptr;      // its levels and work vectors, kept across solves on the same lattice
    struct Mg *mg_slab = nullptr; // the same for a z-slab of a row-sharded lattice (pgd_mg_slab_*: levels >= 1 are whole and replicated)
    int64_t mg_solves = 0, mg_fallbacks = 0;
    int mg_chunk = 2;             // iterations queued between two looks at the flags when the multigrid preconditioner is on (even: the slot parity of a replayed chunk)
    int mg_march_min = 64;        // levels with at least this many nodes along x and y run their stencil passes in k_spmv_stencil_march
    int cls_cache_on = 1;         // classification of an operator whose structure was seen before: codes copied, every row verified (PGD_TUNE_CLS_CACHE)
    int64_t cls_fast = 0, cls_full = 0;      // classifications served by the cache / done in full
    int spmv_stencil = 1;         // ... and its stencil form (couplings in scalar registers, four rows per thread) where every row verifies
    int spmv_zchunk_stencil = 0;  // > 0: planes per march of k_spmv_stencil_march (0: fill every workgroup slot once)
    int stencil_depth = 0;        // plane fetches in flight per workgroup of k_spmv_stencil_march (0: chosen from the march length; 3 or 6)
    int stencil_wg_per_cu = 2;    // resident workgroups per CU of k_spmv_stencil_march (sets the march length)
    int spmv_classes = 1;         // row-class dictionary of the scaled diagonal form (k_spmv_diac_march2) where the operator has one
    void *cls_scratch = nullptr;  // hash slots of dia_classify
    double *gram_w = nullptr;     // pgd_start_gram: the products A v_j, one vector each
    size_t gram_w_bytes = 0;
    pgd_handle gram_op = 0;       // ... of this operator, gram_k of them, each gram_n long (pgd_start_residual reads them; 0: none held)
    int gram_k = 0;
    int64_t gram_n = 0;
    int spmv_variant = 0;         // z-march: 0 = k_spmv_dia_march2 (64 x 8 patch, two rows per thread), 1 = 64 x 8 / 512 threads, 2 = 64 x 4 / 256 threads
    int spmv_zchunk = 8;          // k_spmv_dia_march: most planes a workgroup marches through (0: never use that kernel)
    int pcg_fold_reduce = 1;      // scaled recurrence: final reduction passes folded into the vector kernels (3 launches / iteration)
    int spmv_qq = 0;              // transient: the DIA products also leave y . y partial sums (pairs per workgroup)
    int pcg_single_sync = 1;      // scaled recurrence on structured grids above 2^20 rows: one reduction + one vector kernel per iteration
    int spmv_unit_diag = 1;       // scaled recurrence on structured grids: the unit diagonal is set to exactly 1 and not loaded
    int pcg_defer_x = 1;          // scaled recurrence, large systems: x += alpha p in the p kernel (8 vector passes per iteration, not 9)
    int pcg_scaled = 1;           // pgd_pcg_solve on the symmetrically scaled system (no dinv / z passes) when the symmetric storage applies
    int stencil_rows = 0;         // rows per thread of k_spmv_stencil_march: 0 = by the launch's shape (2 for thin slabs), 2, 4 (PGD_TUNE_STENCIL_ROWS)
    int dia_march3 = 0;           // 1: variant 0 of the plain z-march in k_spmv_dia_march3 (PGD_TUNE_DIA_MARCH3; bit-identical, measured 2 - 5 % slower at 256^3)
    int push_in_update = 1;       // direct halo: the boundary planes leave from the update kernel itself (PGD_TUNE_PUSH_IN_UPDATE); 0: k_halo_push
    int shard_one_march = 1;      // pgd_pcg_solve_sharded, exchange in stream order: all owned planes in one stencil march, ghost planes as data (PGD_TUNE_SHARD_ONE_MARCH)
    int pcg_derive_scaled = 1;    // ... whose stencil couplings are DERIVED from the verified stencil of A where that exists (PGD_TUNE_PCG_DERIVE_SCALED)
    int spmv_combine_dia = 1;     // structured grids: op_combine also forms the diagonal form from the atoms' diagonal forms
    int spmv_sym = 1;             // PCG products from the symmetric half storage when the mesh qualifies

    // SpMV launch timing (HIP events on `stream`)
    bool prof = false;
    bool prof_pcg_only = false;   // time only the PCG instance k_spmv_csr<dot,store>
    std::vector<hipEvent_t> ev;   // pairs
    // per pair: kind 0 = a product launch, 1 = the vector update of the single-sync recurrence; the bytes it is priced with; the
    // iteration of the PCG solve it belongs to (-1: not inside a solve's loop).  Launches queued behind the iteration that
    // converged are no-ops (every kernel returns on the done flag): their samples are dropped when the solve knows how far it
    // got (prof_commit), instead of entering the averages with full bytes and no time.
    struct ProfRec { uint8_t kind; int iter; double bytes, own, seconds; };
    std::vector<ProfRec> ev_rec;
    std::vector<ProfRec> prof_pend;   // measured, waiting for their solve's verdict
    int prof_iter = -1;               // iteration whose launches are being queued (set by the PCG loops)
    int64_t prof_dropped = 0;
    double prof_overhead = -1.0;      // seconds a pair of HIP events adds to the kernel it brackets (prof_calibrate; -1: not measured)
    size_t ev_used = 0;
    int64_t prof_upd_launches = 0, prof_upd_seen = 0;   // k_pcg1_update, timed like the products (one launch in four)
    double prof_upd_seconds = 0.0, prof_upd_bytes = 0.0;
    int64_t prof_launches = 0, prof_seen = 0;
    double prof_seconds = 0.0, prof_bytes = 0.0;
    double prof_own_bytes = 0.0;  // least bytes the kernels that were timed must move in their own storage form
    int64_t kcount[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // launches per product kernel family (KC_*)
    hipEvent_t timer_ev[2] = {nullptr, nullptr};     // pgd_timer_start / pgd_timer_stop
    int *flags_host = nullptr;                       // pinned: two snapshots of the device flags (the PCG loops look at one chunk's
    hipEvent_t flag_ev[2] = {nullptr, nullptr};      //   flags while the next chunk is already queued: pcg_flag_snapshots)
    int pcg_pipeline = 1;                            // 1: the next 16-iteration chunk is queued before the host looks at the flags of the last
};

enum { KC_CSR = 0, KC_CSR_DICT = 1, KC_SYM_ROWS = 2, KC_DIA_ROWS = 3, KC_DIA_MARCH = 4, KC_MULTI = 5, KC_DIAC_MARCH = 6, KC_STENCIL_MARCH = 7 };

// ---- helpers implemented in pgd_ctx.hip
Ctx *get_ctx(pgd_handle h);
int fail(Ctx *c, int code, const char *fmt, ...);
pgd_handle put_obj(Ctx *c, Obj *o);
Obj *get_obj(Ctx *c, pgd_handle h, Obj::Kind k);
int free_obj(Ctx *c, pgd_handle h, Obj::Kind k);
int dev_alloc(Ctx *c, void **p, size_t bytes);
int ensure_partials(Ctx *c, int64_t n);
int ensure_work(Ctx *c, int i, int64_t n);
int ensure_mask(Ctx *c, int64_t n);
int pcg_flag_snapshots(Ctx *c);                      // pgd_ctx.hip: the pinned flag buffers and their events exist
int ensure_ibuf(Ctx *c, int64_t n);
void prof_flush(Ctx *c);
void prof_commit(Ctx *c, int valid_products, int valid_updates);   // end of a solve: samples of iterations that ran are kept
void comm_release(Ctx *c);          // pgd_comm.hip
struct Mesh;
struct Csr;
int build_sym_tables(Ctx *c, Mesh *m);                       // pgd_spmv.hip
int ensure_sym(Ctx *c, const Mesh *m, Csr *a, bool *usable);  // pgd_spmv.hip: convert (once per operator)
int cg_init_s(Ctx *c, const double *b, const double *q, const double *sc, double *r, double *p, double *s, int64_t lo,
              int64_t hi, int base);                                   // pgd_pcg.hip: scaled sharded recurrence
int cg_update_s(Ctx *c, double *x, double *r, const double *w, double *p, double *s, const double *sc, int64_t lo, int64_t hi,
                int base);
int cg_update_s2(Ctx *c, double *x, double *r, const double *w, double *p, double *s, const double *sc, int64_t lo, int64_t hi,
                 int base, int parity, int *nblocks);
int reduce_two_slots(Ctx *c, int na, int nb, int base);
int pcg1_seed(Ctx *c, int npairs, int slot_rz, int slot_rr);          // single-sync recurrence, sharded form (pgd_pcg.hip)
int pcg1_tol(Ctx *c, int base, double rtol, double atol, int slot_p8);
int pcg1_aux(Ctx *c, const double *s, const double *r, int64_t lo, int64_t hi, int slot);
int pcg1_sums(Ctx *c, int nprod, int nvec, int base);
int pcg1_finish_slots(Ctx *c, int base);
// direct halo folded into the update (pgd_comm.hip): the rows [lo, lo_end) and [hi_begin, hi) of the new p also go - as write-through
// stores at system scope - to dst_lo / dst_hi, the workgroups that hold such rows take a ticket, the last one posts `seq` and polls
struct PushArgs {
    double *dst_lo = nullptr, *dst_hi = nullptr;
    int64_t lo_end = 0, hi_begin = 0;
    unsigned long long *post_lo = nullptr, *post_hi = nullptr;
    const unsigned long long *wait_a = nullptr, *wait_b = nullptr;
    unsigned long long seq = 0, *ticket = nullptr;
    unsigned int nblocks = 0;          // filled by pcg1_update: workgroups that take a ticket
    long long ticks = 0;
};
unsigned int pcg1_update_push_blocks(int g, int64_t lo, int64_t hi, int64_t lo_end, int64_t hi_begin);
int pcg1_update(Ctx *c, double *x, double *r, double *p, const double *q, const double *sc, int64_t lo, int64_t hi, int base,
                int *nblocks, int lag, int fold_par, const PushArgs *push = nullptr);
int pcg1_flush_x(Ctx *c, double *x, const double *p, const double *r, int64_t lo, int64_t hi, int base, int fold_par);
int vec_sqrt(Ctx *c, double *v, int64_t n);
int vec_div_mul(Ctx *c, double *x, const double *sc, int64_t n, int mul);
int ensure_vals(Ctx *c, const Mesh *m, Csr *a);                 // pgd_pcg.hip: CSR values of an operator whose combine was deferred
bool atom_fast_form(Ctx *c, const Mesh *m, Csr *a, int64_t r0, int64_t r1);   // pgd_spmv.hip
// pgd_spmv.hip: row-class dictionary of the current slot values (+ its stencil form, verified on the planes [zlo, zhi) - whole grid: -1)
int dia_classify(Ctx *c, const Mesh *m, Csr *a, int zrange_lo = -1, int zrange_hi = -1);
bool stencil_row_range(const Ctx *c, const Mesh *m, const Csr *a, int64_t r0, int64_t r1);   // ... over the whole planes [r0, r1) of a slab
bool stencil_whole_grid(const Ctx *c, const Mesh *m, const Csr *a);   // pgd_spmv.hip: a product over all rows would run in k_spmv_stencil_march
int launch_stencil_pass(Ctx *c, const uint8_t *cls, int ident, const double cst[8], int nx, int ny, int nz, int zm0, int zm1,
                        const double *x, const double *b, double *y, double w, int epi, bool dot, int *nparts, int z0 = 0, int z1 = -1);      // pgd_spmv.hip
// pgd_mg.hip: multigrid preconditioner of the scaled stencil operator
bool mg_prepare(Ctx *c, const Mesh *m, const Csr *a);                          // true: usable for this operator (levels built, buffers there)
int mg_fix_start(Ctx *c, const Csr *a, const double *b, double *x, int64_t n);    // x = b on the eliminated rows
int mg_vcycle(Ctx *c, const double *r, bool dot, int *nparts, double *z_out = nullptr);      // z = M r into z_out (default: mg_result(c)); partial sums of r.z into c->partials
double *mg_result(Ctx *c);
void mg_release(Ctx *c);
int sym_scale(Ctx *c, const Mesh *m, Csr *a, const double *s);   // pgd_spmv.hip: slot values *= s_i s_j
int launch_spmv_dia_rows2(Ctx *c, const Mesh *m, const Csr *a, const double *x, double *y, const double *w, int64_t r0a,
                          int64_t r1a, int64_t r0b, int64_t r1b, bool dot, const int *flags, int *nparts_out, bool *done);
int combine_dia(Ctx *c, const Mesh *m, Csr *o, Csr *const *atoms, const double *coefs, int n, const uint8_t *mask);
int launch_spmv_op(Ctx *c, const Mesh *m, const Csr *a, const double *x, double *y, const double *w, int64_t r0,
                   int64_t r1, bool dot, bool store, const int *flags, int *nparts_out);

inline Vec *get_vec(Ctx *c, pgd_handle h) { return static_cast<Vec *>(get_obj(c, h, Obj::VEC)); }
inline Mesh *get_mesh(Ctx *c, pgd_handle h) { return static_cast<Mesh *>(get_obj(c, h, Obj::MESH)); }
inline Csr *get_csr(Ctx *c, pgd_handle h) { return static_cast<Csr *>(get_obj(c, h, Obj::CSR)); }

#define PGD_CTX(c, h)                         \
    pgd::Ctx *c = pgd::get_ctx(h);            \
    if (!c) return PGD_ERR_INVALID;           \
    (void)hipSetDevice(c->device)

#define PGD_HIP(c, call)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return pgd::fail(c, PGD_ERR_HIP, "%s failed: %s (%s:%d)", #call,                \
                             hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define PGD_TRY(call)                \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != PGD_OK) return rc_; \
    } while (0)

#define PGD_LAUNCH_CHECK(c) PGD_HIP(c, hipGetLastError())

inline int grid_for(int64_t n, int per_block = TPB, int cap = MAX_VEC_BLOCKS) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- launchers implemented across translation units
// pgd_vec.hip
int scan_exclusive_i32(Ctx *c, const int *in, int *out, int64_t n);   // out has n+1 entries
int reduce_partials(Ctx *c, const double *partials, int nparts, int nvals, int slot0, int check_mode,
                    int slot_rr, int slot_tol2);
int reduce_partials_to(Ctx *c, const double *partials, int nparts, int nvals, double *dest);
int k_reduce_stage1_pub(Ctx *c, const double *partials, int nparts, int nvals, double *out);
int vec_dot_range(Ctx *c, const double *x, const double *y, int64_t lo, int64_t hi, int slot);
// pgd_spmv.hip
int launch_spmv(Ctx *c, const Mesh *m, const double *vals, const double *x, double *y,
                const double *w, int64_t r0, int64_t r1, bool dot, bool store, const int *flags,
                int *nparts_out);
int launch_spmv_multi(Ctx *c, const Mesh *m, const double *vals, const double *x,
                      const double *const *ys, int ny, int64_t r0, int64_t r1, double *out_host);
// pgd_pcg.hip
int csr_diag_inv(Ctx *c, const Mesh *m, Csr *a);

}  // namespace pgd

// ---- device helpers --------------------------------------------------------------
namespace pgd {

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Give every XCD one contiguous chunk of the logical tile range so that
// neighbouring row blocks - which gather the same x planes - hit the same L2.
// Bijective for any grid size; placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, rem = nblk & 7, k = bid & 7;
    return k * q + (k < rem ? k : rem) + (bid >> 3);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over an NW-wave workgroup in a fixed order; result valid in thread 0.
template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double *s_red /* >= NW */) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NW; ++k) t += s_red[k];
    }
    return t;
}

// Sum over a 256-thread workgroup in a fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *s_red /* >= 4 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    return (threadIdx.x == 0) ? ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) : 0.0;
}

}  // namespace pgd
