/* =============================================================================
 * csgpu.h -- C ABI of libcsgpu.so: the MI355X (gfx950) Laplacian-solve backend for Circuitscape.
 *
 * This is the drop-in boundary for ONE hot path of the reference (Circuitscape.jl v5.17.1):
 * AMG-preconditioned CG on the SPD graph Laplacian for pairwise effective resistance.
 * Plain C types only (pointers + sizes); no torch / HIP types appear in any signature.
 *
 * Reference interfaces each entry point replaces (paths relative to /root/reference):
 *
 *   csgpu_setup            <-> construct_cholesky_factor(matrix, ::XSolver) -> handle     src/core.jl:519-523,
 *                              ext/CircuitscapePardisoExt.jl:31-32; and the AMG setup site
 *                              aspreconditioner(smoothed_aggregation(matrix; ...))         src/core.jl:164-167
 *   csgpu_solve_rhs        <-> solve_linear_system(factor, matrix, rhs::Matrix) -> lhs     src/core.jl:646-653,
 *                              ext/CircuitscapePardisoExt.jl:34-45; iterative flavour
 *                              solve_linear_system(G, curr::Vector, M)                     src/core.jl:636-643;
 *                              multiple_solve(s, matrix, sources)                          src/raster/advanced.jl:307-312
 *   csgpu_solve_pairs      <-> the per-pair body of solve(prob, ::AMGSolver, ...):
 *                              RHS -1/+1 (core.jl:224-226), solve (:229), grounding shift and resistance
 *                              (:231-232), focal-voltage gather for the shortcut (update_voltmatrix! :685-703);
 *                              batched like the direct-solver driver (core.jl:448-493)
 *   csgpu_solve_region_pairs <-> the per-pair graph + hierarchy of _pt_file_polygons_path (src/raster/pairwise.jl:72-135)
 *   csgpu_solve_grounded   <-> multiple_solver with infinite grounds (src/raster/advanced.jl:274-305) as the one-to-all /
 *                              all-to-one drivers call it per focal point (src/raster/onetoall.jl:106-151): many
 *                              ground sets, one hierarchy
 *   csgpu_solve_sources    <-> the same for the sparse right-hand sides those drivers build (one +1 per one-to-all column), with
 *                              what they keep of a solve: `res[i] = v[1]` (onetoall.jl:141) and the accumulated current maps
 *                              (onetoall.jl:153-158); network advanced mode src/network/advanced.jl:1-51
 *   csgpu_multi_solve_grounded, csgpu_multi_solve_sources
 *                          <-> the fan-out of those drivers over the focal points (Threads.@spawn per point,
 *                              src/raster/onetoall.jl:146-151) as one host thread per GPU
 *   csgpu_solve_pairs_currents <-> the same plus postprocess() -> write_cur_maps -> _create_current_maps
 *                              (core.jl:655-683, out.jl:46-115,150-303): node currents, cumulative and maximum maps
 *   csgpu_multi_setup, csgpu_multi_raster_setup, csgpu_multi_solve_pairs, csgpu_multi_solve_pairs_currents, csgpu_multi_free
 *                          <-> the task fan-out and serial result merge of solve(prob, ::AMGSolver, ...)
 *                              (Threads.@spawn per source point, src/core.jl:262-285), as one host thread per GPU
 *   csgpu_free             <-> GC finalizer of the factor object (PardisoFactorize, Pardiso ext :8-13)
 *   csgpu_last_error       <-> error(msg) strings (core.jl:641,650)
 *   csgpu_raster_setup     <-> construct_node_map/construct_graph/laplacian! for a raster without polygons (NODATA allowed)
 *                              (src/raster/pairwise.jl:271-362, src/core.jl:608-634) -- "next" row N4, used by
 *                              bench.py so the synthetic Laplacian is born in HBM
 *   csgpu_raster_setup_poly <-> the same with short-circuit polygons (node merging, summed parallel edges;
 *                              src/raster/pairwise.jl:276-301, 316-362)
 *   csgpu_raster_nodemap   <-> the node map construct_node_map returns (src/raster/pairwise.jl:271-301)
 *   csgpu_components       <-> connected_components(SimpleGraph(G))       src/raster/pairwise.jl:233,
 *                              src/raster/advanced.jl:59, src/network/pairwise.jl:52
 *   csgpu_raster_setup_grounded, csgpu_solve_raster
 *                          <-> compute_omniscape_current(conductance, source, ground, cfg)  src/utils.jl:145-257 and
 *                              the raster branch of advanced_kernel (src/raster/advanced.jl:151-271) for rasters
 *                              without polygons: grounded matrix, right-hand side from the source raster, node
 *                              currents incl. ground currents (src/out.jl:178-207), maps back -- "next" rows N2/N3
 *   csgpu_get_info, csgpu_spmv_bench, csgpu_spmv_host, csgpu_level_spmv_host, csgpu_get_level_matrix,
 *   csgpu_dia_product_host <-> no reference counterpart: measurement and test hooks
 *
 * Conventions
 *   - The matrix is a symmetric SPD (or singular-consistent) graph Laplacian in compressed sparse
 *     column/row form (identical by symmetry) exactly as Julia's SparseMatrixCSC stores it:
 *     rowptr[n+1], colidx[nnz] (sorted within a row), vals[nnz]; idx_bytes in {4,8}, val_bytes in {4,8},
 *     index_base in {0,1}. The library COPIES it to the device; host pointers are never retained.
 *   - On device everything is int32 / 0-based and n < 2^31 - 1 is required (status 4 otherwise); vector element indices
 *     (node * batch + column) are 64-bit. A device CSR form exists only for nnz < 2^31. csgpu_raster_setup builds the fine
 *     level of a raster without a CSR matrix, so rasters are limited by n = rows * cols < 2^31 - 1 only (tested at
 *     21000 x 21000 = 441 M cells, 3.97e9 stored entries). A HOST matrix with nnz >= 2^31 (the reference's
 *     use_64bit_indexing, src/run.jl:34: Int64 colptr / rowval) is accepted by csgpu_setup when it is the graph of a raster
 *     without polygons and comes with csgpu_opts.node_row / node_col: it is streamed to the device in blocks of rows and
 *     scattered straight into the same lattice form (csgpu_get_info().host_blocks tells; status 4 with the reason when the
 *     matrix does not qualify). Calls that need the CSR form of such a handle (voltage / current maps, explicit_check)
 *     return status 4, resistance-only csgpu_solve_pairs does not.
 *   - All calls are blocking; a handle is serialised internally (one caller at a time per handle).
 *   - Return value: 0 ok, 1 not converged (some right-hand side failed the reference's 1e-4 true-residual check,
 *     src/core.jl:640-641 -- the reference's only acceptance test: how the iteration stopped (rule met, itmax, breakdown of
 *     the recurrence) is not looked at there and does not decide here), 2 HIP runtime error, 3 out of memory, 4 bad arguments,
 *     5 internal error.
 *     csgpu_last_error() returns a thread-local human-readable message for the last non-zero status.
 * ============================================================================= */
#ifndef CSGPU_H
#define CSGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct csgpu_handle csgpu_handle;

enum {
  CSGPU_OK = 0,
  CSGPU_NOT_CONVERGED = 1,
  CSGPU_HIP_ERROR = 2,
  CSGPU_OOM = 3,
  CSGPU_BAD_ARGS = 4,
  CSGPU_INTERNAL = 5
};

/* Convergence rules. 0 = Krylov.jl rule used by the reference (core.jl:639): stop when
 * sqrt(r'M^-1 r) <= atol + rtol*sqrt(r0'M^-1 r0). 1 = true residual: ||r||_2 <= atol + rtol*||b||_2. */
/* BOTH: the reference's rule and the true-residual rule must hold. Solves with Dirichlet sets on a shared hierarchy
 * (csgpu_solve_grounded, csgpu_solve_region_pairs) run BOTH when KRYLOV is asked for: the preconditioned norm of the
 * ungrounded hierarchy carries a near-kernel gain the reference's own grounded hierarchy does not have (csrc/pcg.h). */
enum { CSGPU_CRIT_KRYLOV = 0, CSGPU_CRIT_TRUE_RESIDUAL = 1, CSGPU_CRIT_BOTH = 2 };

/* Aggregation strategy. AUTO = grid tiles when node coordinates are supplied, else MIS(2). */
enum { CSGPU_AGG_AUTO = 0, CSGPU_AGG_MIS2 = 1, CSGPU_AGG_GRID = 2 };

typedef struct csgpu_opts {
  int32_t struct_size;    /* = sizeof(csgpu_opts); set by csgpu_default_opts */
  int32_t device;         /* HIP device ordinal, -1 = current device */
  int32_t max_levels;     /* default 16 */
  int32_t max_coarse;     /* stop coarsening at <= this many unknowns (dense pseudo-inverse); default 100 */
  int32_t aggregation;    /* CSGPU_AGG_* */
  int32_t nu_pre;         /* damped-Jacobi pre-smoothing sweeps, default 1 */
  int32_t nu_post;        /* damped-Jacobi post-smoothing sweeps, default 1 */
  int32_t criterion;      /* CSGPU_CRIT_*, default KRYLOV (the reference's rule) */
  int32_t itmax;          /* default 100000 (core.jl:639) */
  int32_t batch;          /* right-hand sides solved together per SpMM pass: 1,2,4,8,16,32; default 8. 32 is what the
                             marching kernels of a raster want (+8 % fp64 / +10 % mixed over 16 at 10000^2); fp64
                             hierarchies without a lattice level 0 (networks, thin-polygon rasters) are held at 16 */
  int32_t check_every;    /* host polls the device convergence flags every this many iterations; default 0 = auto:
                             every iteration when n*batch >= 2^25 (an iteration then takes milliseconds), else every 4th */
  int32_t nu_coarse;      /* Jacobi sweeps (pre and post) on level 1; the levels below it (1/81 of the fine level's
                             work) run one more (profiles/r2_sweeps_per_level.json). On these levels the sweeps of a
                             level carry Chebyshev weights (the reciprocal roots of the Chebyshev polynomial of degree
                             = sweeps on [rho/10, rho], rho = Gershgorin bound; profiles/r2_coarse_chebyshev.json) unless
                             the hierarchy is fp32 above 3e7 rows, which keeps one damped weight. Default 2 (measured
                             on the 10000^2 raster: 3 -> 324.7 ms per batch of 16 at 12.8 iterations, 2 -> 309.2 ms at
                             12.9, 1 -> 327.4 ms at 14.9; profiles/r2_polling_graph_nucoarse.json) */
  double theta;           /* symmetric strength threshold (AlgebraicMultigrid SymmetricStrength) of the MIS(2) aggregation,
                             default 0 (the reference's). Raster lattices keep their regular 3x3 tiles and apply a
                             strength filter of their own (0.03 sqrt(a_ii a_jj), only on heterogeneous rasters: csrc/amg_setup.h,
                             TileStrength); theta != 0 switches the lattice paths off */
  double omega_p;         /* prolongator smoothing weight over local row-abs-sum weighting: P = T - omega_p Dl^-1 A T.
                             The reference's JacobiProlongation uses 4/3; 1.6 (default) measured 35 % fewer PCG
                             iterations on the 10000^2 raster (tools/sweep.sh) */
  double omega_s;         /* Jacobi smoother weight numerator: omega = omega_s / rho_gershgorin (< 2/rho: always
                             convergent), default 1.7: measured better than or equal to 1.5 on 8- and
                             4-neighbour, homogeneous, log-normal, NODATA and averaged-resistance rasters */
  double rtol;            /* default 1e-6 (core.jl:639) */
  double atol;            /* < 0 means sqrt(eps(T)) (Krylov.jl default); default -1 */
  /* Optional raster coordinates of every node (length n, 0-based cell row / col of the node's first
   * cell). When given, aggregation seeds 3x3 tiles (the shape the reference's greedy StandardAggregation
   * produces on rasters). NULL for network graphs. With coordinates a raster graph WITH NODATA cells (every node on a
   * cell of its own, couplings between neighbouring cells only, at least half of the bounding box valid) is scattered
   * into the lattice of its bounding box and runs the index-free kernels exactly like a raster handed to
   * csgpu_raster_setup (setup_cellspace_from_csr, csrc/csgpu.hip); node ids and n-vectors at this boundary keep the
   * caller's numbering. An all-valid raster needs no coordinates for that: its lattice period is read off the matrix. */
  const int32_t* node_row;
  const int32_t* node_col;
  /* Storage / arithmetic precision of the AMG preconditioner (hierarchy + V-cycle): 0 = same as val_bytes,
   * 4 = fp32 preconditioner under an fp64 CG iteration (residuals, search directions, dot products and the
   * reference's residual check stay in val_bytes precision). Default 0. A problem that is not coarsened
   * (n <= max_coarse: the preconditioner is the dense pseudo-inverse) computes in val_bytes precision whatever this says. */
  int32_t precond_bytes;
  /* Replay the PCG iteration as a captured hipGraph of check_every iterations between two host polls:
   * 0 = auto (on when n*batch <= 2^25, the launch-latency-bound regime), 1 = always, -1 = never. Default 0. */
  int32_t use_graph;
  /* Level 0 of a V(1,1) cycle as two products, b_c = Q^T b and out = [S Q][b; x_c], instead
   * of residual + restriction + fused prolongation: 0 = on whenever nu_pre == nu_post == 1 (default), -1 = off. */
  int32_t two_product;
  /* Lattice ("symmetric diagonal") form of the fine-level matrix for the CG product (csrc/stencil.h): detected from
   * the matrix itself -- an all-valid raster in the reference's column-major numbering couples node i to i+-1,
   * i+-(R-1), i+-R, i+-(R+1) only -- and used with the search-direction update fused in. 0 = use it when the
   * matrix has that form (default), -1 = always the CSR product. */
  int32_t stencil;
  /* Resistance-only pair solves (csgpu_solve_pairs without volt_out) accumulate the solution at the focal nodes only
   * and evaluate the reference's 1e-4 post-check (core.jl:640) on the fp64 recurrence residual. 1 = carry the whole
   * solution vector and evaluate ||A x - b|| / ||b|| with an explicit product, as every other entry point does.
   * Default 0. */
  int32_t explicit_check;
  int32_t reserved3;
  /* ---- Round 6: every decision that used to be an environment switch inside the library. 0 = the
   * library's default in every field; csgpu_default_opts writes zeros. The CSGPU_* environment variables that carried these
   * choices remain as DEBUG overrides only: they are read ONCE, when a handle is set up (csrc/csgpu.hip: knobs_from_opts),
   * never on a call path, and a handle keeps what it was set up with -- two handles of one process may differ.
   * csgpu_get_info reports what a handle ended up with (level_form, enrich_vectors, hierarchy_rebuilt_fp64, host_blocks,
   * batch_width, stream_mode, tail_first_level, last_level_sweeps, coarse_chebyshev, cellspace, poly_lattice, expander_probe_hit). */
  int32_t last_level_sweeps;  /* damped-Jacobi sweeps that stand in for the coarsest solve when the last level is too large
                                 for a dense inverse. A hierarchy of ONE level -- a graph the set-up declines to coarsen,
                                 BASELINE configs[4] -- is then CG with a polynomial preconditioner of that degree: every sweep
                                 is one more pass over the matrix per iteration. 0 = default: 8 below a coarsened hierarchy,
                                 1 for a single-level one (measured on the 5e6-node network, profiles/r6_network_sweeps.jsonl);
                                 -1 = none (plain Jacobi scaling) */
  int32_t enrich;             /* second coarse function on badly shaped aggregates (csrc/enrich.h): 0 = on where the tiles
                                 were refined, -1 = off */
  int32_t enrich_steps;       /* power steps of the local Fiedler vectors, 0 = 6 */
  int32_t dia25_min_rows;     /* smallest coarse level that takes the 25-point lattice form (csrc/dia25.h): 0 = 16384, -1 = none */
  int32_t dia25_prefetch;     /* 0 = b and D^-1 loaded one column ahead (default), -1 = off */
  int32_t dia25_waves;        /* waves per SIMD asked of the 25-point kernel, 0 = the library's rule */
  int32_t dia25_fused_j0;     /* 0 = first two sweeps of a 25-point level as one marching pass (default), -1 = off */
  int32_t stream;             /* streaming pair solves (a column takes the next pair when its own has converged): 0 = decided
                                 by the spread of the first batch's iteration counts, 1 = from the first pair on, -1 = never */
  int32_t tail_rows;          /* coarse levels with at most this many rows run in ONE launch (csrc/tail.h): 0 = 4096, -1 = off */
  int32_t poly_lattice;       /* csgpu_raster_setup_poly on the lattice path: 0 = when every polygon is contiguous and none
                                 is long and thin, 1 = whatever the shapes, -1 = never (merged CSR graph) */
  int32_t cellspace;          /* rasters with NODATA cells on the full lattice: 0 = when at least cellspace_min_frac of the
                                 cells are valid, -1 = never (compact CSR numbering) */
  int32_t cellspace_from_csr; /* the same for a host CSR matrix that comes with node_row / node_col: 0 = auto, -1 = never */
  int32_t lattice_level1;     /* level 1 of an all-valid raster in the collapsed four-product lattice form: 0 = auto, -1 = off */
  int32_t lattice_level1_min_rows; /* 0 = 16384 */
  int32_t lattice_setup;      /* level 0 built from the raster without a CSR matrix (csrc/lattice_setup.h): 0 = auto, -1 = off */
  int32_t lattice_s;          /* S of the two-product level in lattice form: 0 = auto, -1 = CSR */
  int32_t lattice_q;          /* Q of the two-product level in its index-free form: 0 = auto, -1 = CSR */
  int32_t direct_tiles;       /* 3x3 tiles written down directly on rasters of known extent: 0 = yes, -1 = MIS(2) seeded */
  int32_t tile_pieces;        /* per-tile piece analysis (NODATA lines, weak couplings): 0 = on, -1 = off */
  int32_t direct_at;          /* A*T by the one-thread-per-row kernel: 0 = on, -1 = general SpGEMM */
  int32_t dirichlet_coarse;   /* coarsest-level correction of Dirichlet-masked solves: 0 = on, -1 = off */
  int32_t deflation;          /* near-kernel eigenpair of an fp32 hierarchy's coarsest operator dropped: 0 = on, -1 = off */
  int32_t tail_projection;    /* candidate projected out of the coarse tail's right-hand sides (fp32): 0 = on, -1 = off */
  int32_t coarse_smoother;    /* sweeps of levels >= 1: 0 = Chebyshev weights unless the hierarchy is fp32 above 3e7 rows,
                                 1 = Chebyshev weights always, 2 = one damped-Jacobi weight */
  int32_t nu_l1;              /* sweeps on level 1, 0 = nu_coarse */
  int32_t nu_deep;            /* sweeps below level 1, 0 = nu_coarse + 1 */
  int32_t wide_csr;           /* 1 = fp64 hierarchies without a lattice level 0 run batches of 32 too */
  int32_t fixed_k;            /* 1 = every batch of a call runs at the call's width (the short last batch is padded with
                                 idle columns: round-5 behaviour; default 0 = the width is picked per batch) */
  int32_t recompute_ap;       /* lattice path: 0 = the residual update recomputes A p (default), -1 = A p stored and re-read */
  int32_t longrow;            /* long-row CSR kernel for R / Q^T: 0 = on, -1 = off */
  int32_t narrow_tile;        /* 1 = the narrow SpMM tile for [S Q] */
  int32_t spmv_grid_cap;      /* workgroups per CSR product, 0 = 65536 */
  int32_t dia_seg;            /* raster columns per tile of the marching kernels, 0 = 32 (64 at K = 32) */
  int32_t restrict_seg;       /* coarse columns per tile of the marching restriction, 0 = 32 (64 in the fused pass) */
  int32_t collapse_min;       /* partial rows above which the dot partials are collapsed first, 0 = the library's rule */
  int32_t verbose;            /* 1 = one line per set-up decision on stderr */
  int32_t expander_probe;     /* large graphs without coordinates: 0 = before the MIS(2) aggregation of level 0 a sample of
                                 2-hop balls predicts nnz(P) / nnz(A); above 0.8 the graph is an expander whose aggregation the
                                 set-up would throw away (nnz(P) > 0.75 nnz(A)) and the handle gets its one level at once
                                 (BASELINE configs[4]: 0.20 -> 0.03 s of device set-up); -1 = always aggregate first */
  int32_t fused_restrict;     /* lattice path, batches of 16 / 32 columns in one precision, resistance-only pair solves: 1 = the
                                 residual update and the restriction of the V-cycle run as ONE marching pass over r (the
                                 residual ping-pongs between two buffers: + n x batch values of device memory; results are
                                 those of the two-pass path bit for bit); -1 = two passes; 0 = fused in double precision
                                 (+7 % pair-solves/s at 10000^2), two passes in single precision (where the fused pass is
                                 3 - 5 % slower). Levels with enriched aggregates (rasters with
                                 NODATA cells, csrc/enrich.h) take the fused pass too: the enrichment's change of the residual
                                 reaches b_c through a coarse-side correction (W = Q'AE), whose sums run in another order --
                                 THERE the fused and the two-pass results agree to rounding, not bit for bit */
  int32_t sparse_init;        /* fused lattice path, pair solves: 0 / 1 = the right-hand side of a batch, r0 = e_dst - e_src, is
                                 never stored -- the first restriction scatters <= 18 entries per column, the first second
                                 product and the first residual update synthesise it (three passes over n x batch values and
                                 the clearing of r saved per batch, same bits); -1 = r0 written and read like any residual */
  int32_t fused_level1;       /* lattice V(2,2) levels (level 1 of a full raster): 1 = x = S b and b_c = Q2' b in one marching pass
                                 over b, -1 = two passes, 0 = fused in double precision; same bits either way */
  int64_t stream_min;         /* vector elements n * batch from which streaming is considered, 0 = 2^25 */
  int64_t host_stream_block;  /* csgpu_setup: stream the host matrix in blocks of at most this many entries (test / tuning);
                                 0 = only matrices with >= 2^31 stored entries, in blocks of 2^28 */
  double enrich_tau;          /* local Fiedler value below which an aggregate gets its second function, 0 = 0.06 */
  double hetero_fp64_frac;    /* an fp32 hierarchy is rebuilt in fp64 when more than this fraction of the cells leaves its
                                 tile in the strength test, 0 = 0.03, >= 1 = never */
  double poly_strength;       /* interior strength of every polygon (poly.h), 0 = per polygon from its size and shape */
  double poly_coef, poly_smin, poly_smax; /* ... = clamp(coef * cells * blob share, smin, smax); 0 = 1, 8, 1000 */
  double cellspace_min_frac;  /* 0 = 0.5 */
  double tile_theta;          /* strength filter of the tiles, 0 = 0.03 (negative: off) */
  double tile_split_min;      /* ... applied when more than this fraction of the cells would leave their tile, 0 = 0.005 */
} csgpu_opts;

/* csgpu_info.level_form: CSR = general CSR SpMM; LATTICE9 = index-free nine-point lattice form (level 0: the marching
 * CG product / two-product level, csrc/stencil.h + lattice.h; level 1: the collapsed four-product form of
 * lattice_level1_setup); LATTICE25 = A in the 25-point lattice form of csrc/dia25.h (P, R, Q stay CSR; reported for handles
 * whose full batches are at least 8 columns wide -- narrower calls run such a level through the CSR SpMM); TAIL = the level
 * runs inside the single-launch coarse tail (csrc/tail.h; the last level there is the dense pseudo-inverse). */
enum { CSGPU_FORM_CSR = 0, CSGPU_FORM_LATTICE9 = 1, CSGPU_FORM_LATTICE25 = 2, CSGPU_FORM_TAIL = 3 };

typedef struct csgpu_info {
  int64_t n;
  int64_t nnz;
  int32_t levels;               /* including the coarsest */
  int32_t val_bytes;
  int32_t precond_bytes;
  int32_t lattice_period;       /* raster height R when the CG product runs from the lattice form, else 0 */
  double operator_complexity;   /* sum_l nnz(A_l) / nnz(A_0) */
  double grid_complexity;       /* sum_l n_l / n_0 */
  double setup_ms;              /* device time of the AMG setup (HIP events) */
  double upload_ms;             /* host->device copy + index conversion */
  int64_t device_bytes;         /* bytes held by the handle */
  int64_t level_n[32];
  int64_t level_nnz[32];
  int64_t spmv_bytes_fine;      /* algorithmic bytes of one fine-level CSR SpMV (SURVEY.md 8d formula) */
  int64_t bytes_per_iteration;  /* algorithmic bytes of one PCG iteration at batch 1 */
  int32_t level_form[32];       /* CSGPU_FORM_* of every level: which kernels the V-cycle runs the level's products with A
                                   through (tests and A/B scripts read the path a handle took from here, not from side
                                   effects such as device_bytes) */
  int32_t hierarchy_rebuilt_fp64; /* 1: an fp32 hierarchy was asked for (precond_bytes = 4) and the setup replaced it by an
                                   fp64 one (strongly heterogeneous raster / off-diagonal contrast above 1e5 in a host CSR) */
  int32_t enrich_vectors;       /* aggregates of level 0 that carry a SECOND coarse function (csrc/enrich.h: badly shaped aggregates
                                   of a raster with NODATA cells / strength-refined tiles); 0 = none */
  int32_t host_blocks;          /* > 0: the host matrix of csgpu_setup was streamed to the device in this many blocks of rows and
                                   scattered straight into the lattice form (matrices with 2^31 stored entries and more; the
                                   device holds no CSR form of it); 0 = any other set-up */
  int32_t reserved_info;
  /* round 6: the choices a handle ended up with (options, defaults and debug overrides resolved at set-up) */
  int32_t batch_width;          /* columns of a full batch (opts.batch after the K <= 16 rule of fp64 CSR-path handles) */
  int32_t stream_mode;          /* csgpu_opts.stream as resolved: 0 auto, 1 always, -1 never */
  int32_t tail_first_level;     /* first level inside the single-launch coarse tail, -1 = no tail */
  int32_t last_level_sweeps;    /* Jacobi sweeps of a last level without a dense inverse (0 = scaling only); -1 = dense inverse */
  int32_t coarse_chebyshev;     /* 1 = Chebyshev weights on the coarse levels, 0 = one damped-Jacobi weight */
  int32_t cellspace;            /* 1 = one device row per raster CELL (NODATA rasters on the lattice kernels) */
  int32_t poly_lattice;         /* 1 = polygon raster on the lattice path (projected PCG), 0 = merged CSR graph / no polygons */
  int32_t enrich_on;            /* 1 = the enrichment was allowed (enrich_vectors tells how many aggregates took it) */
  double enrich_tau;            /* threshold in effect */
  int32_t expander_probe_hit;   /* 1 = the expansion probe predicted the expander bail-out and the aggregation was skipped */
  int32_t fused_restrict_solves; /* batches so far whose PCG ran the fused residual update + restriction (csgpu_opts.fused_restrict) */
  int32_t virtual_rhs_solves;    /* ... of which the right-hand side was never stored (csgpu_opts.sparse_init) */
  int32_t reserved_info3;
} csgpu_info;

typedef struct csgpu_stats {
  int32_t nrhs;
  int32_t max_iters;            /* max over right-hand sides */
  int64_t total_iters;          /* sum over right-hand sides */
  double max_relres;            /* max over rhs of ||A x - b|| / ||b|| (the reference's post-check, core.jl:640) */
  double solve_ms;              /* wall time of the call, host clock */
  double device_ms;             /* HIP-event time of the PCG loops only */
  double cg_spmv_ms;            /* sum of HIP-event durations of the fine-level CG SpMV/SpMM launches */
  int64_t cg_spmv_calls;        /* number of those launches */
  int32_t batch;                /* batch width actually used */
  int32_t not_converged;        /* number of rhs whose ||Ax-b||/||b|| is not below 1e-4 (core.jl:640), whatever stopped them */
  int64_t graph_launches;       /* hipGraph replays issued (each = check_every PCG iterations) */
  int64_t polished_batches;     /* batches that were re-opened on the true residual because a column stopped on the
                                   configured rule with ||Ax-b||/||b|| >= 1e-4 (the reference would have errored) */
  int64_t cg_spmv_bytes;        /* algorithmic bytes of ONE of the launches timed in cg_spmv_ms:
                                   CSR product: nnz*(val+4) + (n+1)*4 + n*K*(x + val);  lattice product with the fused
                                   search-direction update: n*5*val + n*K*3*x, plus n*K*val when it also stores A p */
  int64_t stream_slots;         /* iterations of the K-wide stream when the call ran as a STREAMING solve (csrc/pcg.h,
                                   pcg_stream_pairs: a column takes the next pair of the list as soon as its own has
                                   converged; a pair costs its own iterations + 1 slots). 0: the batch path. The columns'
                                   utilisation of a call is (total_iters + nrhs) / (stream_slots * batch) */
  /* round 6: the OTHER big launch of an iteration on the lattice path, timed over the same iterations as cg_spmv_ms */
  double resid_ms;              /* sum of HIP-event durations of the residual-update launches (batch path) */
  int64_t resid_calls;          /* number of those launches */
  int64_t resid_bytes;          /* algorithmic bytes of ONE of them: n*5*val + n*K*(p + 2 r [+ the copy of r in the
                                   preconditioner's precision]); fused with the restriction (resid_fused): + n*9*val +
                                   n_coarse*K*val. 0 when the update is not the lattice kernel */
  int32_t resid_fused;          /* 1 = those launches are the fused residual update + restriction (csgpu_opts.fused_restrict) */
  int32_t reserved_stats;
} csgpu_stats;

int csgpu_device_count(void);
void csgpu_default_opts(csgpu_opts* opts);

/* Copy a host CSR/CSC symmetric matrix to the device and build the AMG hierarchy there (construct_cholesky_factor's place,
 * src/core.jl:519-523; the AMG set-up site src/core.jl:164-167). nnz >= 2^31: see "Conventions" above (streamed set-up). */
int csgpu_setup(const void* rowptr, const void* colidx, const void* vals, int64_t n, int64_t nnz, int idx_bytes,
                int val_bytes, int index_base, const csgpu_opts* opts, csgpu_handle** out);

/* Build the 4/8-neighbour Laplacian of a conductance raster (no polygons) directly in HBM and set up AMG.
 * Cells with conductance <= 0 are NODATA (no node), as in construct_node_map. (A raster that is at least half valid keeps
 * one device row per CELL so that the index-free lattice kernels apply -- "cell space"; node ids, n and
 * every n-vector at this boundary stay in the reference's compact numbering.)
 * cond: host pointer, nrows*ncols values (row-major, the orientation of the reference's cellmap[i,j]),
 * all > 0; node numbering is column-major like construct_node_map (raster/pairwise.jl:273-275).
 * Round 5: on such rasters (and on all-valid ones whose tiles the strength test refined) the aggregates whose shape the
 * NODATA cells spoil -- C- / U-shapes around short walls of NODATA cells, hanging together through one neck cell -- carry a
 * SECOND coarse function, applied as a symmetric multiplicative correction around the V-cycle (csrc/enrich.h; 10000^2 with
 * 15 % NODATA: 14.7 -> 13.3 iterations over five masks; csgpu_get_info().enrich_vectors tells how many; CSGPU_ENRICH=0 is the
 * A/B knob). Results are unchanged to the solve tolerance: only the preconditioner differs.
 * reg != 0 applies the reference's regularisation nzval .+= eps(T)*norm(nzval) (core.jl:161). On a raster with several
 * connected components the norm is taken over the WHOLE raster's nonzeros (one handle serves all components), where
 * the reference shifts each component's matrix with that component's own norm (core.jl:158-161): the shifts differ by
 * O(eps), far below the solve tolerance, but the matrices are not bit-identical to the reference's per-component ones. */
int csgpu_raster_setup(const void* cond, int64_t nrows, int64_t ncols, int val_bytes, int four_neighbors,
                       int avg_resistances, int reg, const csgpu_opts* opts, csgpu_handle** out);

/* csgpu_raster_setup for a raster WITH short-circuit polygons (construct_node_map with a polymap,
 * src/raster/pairwise.jl:276-301; known answers test/internal.jl:44-175): polymap[i*ncols + j] > 0 names the polygon of
 * cell (i, j), 0 = none (host pointer, int32, same orientation as cond; NULL = no polygons). Every cell of a polygon --
 * NODATA cells included, as in the reference -- shares the node of the polygon's first valid cell in column-major
 * order; parallel edges are summed, edges inside a polygon vanish. Node numbering, merge and the CSR Laplacian are
 * produced on the device (csrc/raster.h); polygon ids must be < 2^26.
 * Round 4: when every polygon is contiguous and none is long and thin (>= 8 cells long with < 30 % core cells), the handle
 * keeps the raster on the index-free lattice kernels instead (csrc/poly.h): a merged polygon is an equipotential, so PCG runs
 * in the subspace of the vectors that are constant on every polygon (r and z are averaged over each polygon's cells after
 * every update) and the polygon-interior edges, which carry no current for such vectors, are strengthened in the
 * preconditioner's matrix. Same node numbering (csgpu_raster_nodemap is identical), same resistances (1e-7 of a direct
 * solve of the merged matrix at tight tolerances, tools/fuzz_polygons.py); 5000^2 with 50 polygons: 1.22x the polygon-free
 * time per batch instead of 2.25x. csgpu_get_info().lattice_period > 0 tells which path a handle took; everything but
 * resistance-only csgpu_solve_pairs (voltages, currents, general right-hand sides, components, the test hooks) is served by
 * the merged graph, built on first use. CSGPU_NO_POLY_LATTICE=1 forces the merged graph. */
int csgpu_raster_setup_poly(const void* cond, const int32_t* polymap, int64_t nrows, int64_t ncols, int val_bytes,
                            int four_neighbors, int avg_resistances, int reg, const csgpu_opts* opts,
                            csgpu_handle** out);

/* csgpu_raster_setup with a raster of finite ground conductances added to the diagonal (advanced mode:
 * `asolve = a + spdiagm(finitegrounds)`, src/raster/advanced.jl:277-280; `ground` NULL = none). The source / ground
 * conflict policy (remove_src_or_gnd) and direct (infinite) grounds are the caller's business: it applies them to the
 * rasters it hands over (circuitscape.jl_amd/solver.py::raster_advanced_on_device shows how a direct ground becomes
 * a NODATA cell plus ground conductance on its neighbours). */
int csgpu_raster_setup_grounded(const void* cond, const void* ground, int64_t nrows, int64_t ncols, int val_bytes,
                                int four_neighbors, int avg_resistances, int reg, const csgpu_opts* opts,
                                csgpu_handle** out);

/* Advanced-mode solve with rasters in and out on a handle built by csgpu_raster_setup[_grounded]
 * (compute_omniscape_current, src/utils.jl:145-257; advanced_kernel, src/raster/advanced.jl:151-271): the source
 * raster becomes the right-hand side on the device; components without a source or without a ground are skipped as in
 * the reference (advanced.jl:186-191); one PCG solves all components at once (block-diagonal system -- many moving
 * windows can be stacked into one raster, separated by NODATA rows); node currents including the current through the
 * nodes' own ground conductances (src/out.jl:178-207) and/or voltages come back as rasters (row-major, 0 where the cell
 * has no node). curr_out / volt_out may be NULL. */
int csgpu_solve_raster(csgpu_handle* h, const void* source, void* curr_out, void* volt_out, csgpu_stats* stats);

/* Node map of a handle built by csgpu_raster_setup: nodemap_out[i*ncols + j] = 1-based node id of cell (i, j) in the
 * reference's column-major numbering (construct_node_map, src/raster/pairwise.jl:271-301), 0 where the cell is NODATA
 * (conductance <= 0). Pass NULL to query the raster size only. */
int csgpu_raster_nodemap(csgpu_handle* h, int32_t* nodemap_out, int64_t* nrows, int64_t* ncols);

/* Connected components of the handle's graph (connected_components(SimpleGraph(G)), src/raster/pairwise.jl:233,
 * src/raster/advanced.jl:59), computed on the device: component_out[node] (n entries, may be NULL) = dense 0-based
 * component index, components ordered by their smallest node id; *ncomponents = their number. */
int csgpu_components(csgpu_handle* h, int32_t* component_out, int64_t* ncomponents);

int csgpu_get_info(const csgpu_handle* h, csgpu_info* info);

/* Solve A v = e_dst - e_src for every pair (0-based node ids), batched `opts.batch` at a time.
 *   resist_out[p]              = v[dst_p] - v[src_p]                                  (core.jl:232), may be NULL
 *   gathered_out[p*ngather+g]  = v[gather_idx[g]] - v[src_p]                         (may be NULL / ngather 0)
 *   volt_out[p*n + i]          = v[i] - v[src_p]  (column-major n x npairs, core.jl:231), may be NULL
 * All outputs are host pointers of the handle's value type (float or double). */
int csgpu_solve_pairs(csgpu_handle* h, const int64_t* src, const int64_t* dst, int64_t npairs, void* volt_out,
                      const int64_t* gather_idx, int64_t ngather, void* gathered_out, void* resist_out,
                      csgpu_stats* stats);

/* Scope row N1 -- pair solves plus the reference's per-pair current post-processing on the device
 * (get_node_currents src/out.jl:178-207, branch currents :250-290, cumulative / maximum maps :96-107):
 *   curr_out[p*n + i]      = node current of node i for pair p  (max of the current entering and leaving the node,
 *                            branch currents below 1e-8 of the largest one dropped)          may be NULL
 *   cum_curr_inout[i]     += sum_p weights[p] * curr_p[i]       (weights NULL => 1)            may be NULL
 *   max_curr_inout[i]      = max(max_curr_inout[i], max_p curr_p[i])                          may be NULL
 *   branch_out[p*nnz + k]  = |g_k (v_row - v_col)| for stored entry k with row < col (network branch currents,
 *                            out.jl:209-290), 0 at all other positions of the CSR arrays          may be NULL
 *   volt_out, resist_out as in csgpu_solve_pairs. All arrays are host pointers of the handle's value type. */
int csgpu_solve_pairs_currents(csgpu_handle* h, const int64_t* src, const int64_t* dst, int64_t npairs,
                               const int32_t* weights, void* volt_out, void* curr_out, void* cum_curr_inout,
                               void* max_curr_inout, void* branch_out, void* resist_out, csgpu_stats* stats);

/* General right-hand sides: rhs and x_out are host column-major n x nrhs arrays of the handle's value type. */
int csgpu_solve_rhs(csgpu_handle* h, const void* rhs, int64_t nrhs, void* x_out, csgpu_stats* stats);

/* Scope row N2 -- right-hand sides whose systems differ only in WHICH nodes are tied directly to ground, solved on ONE
 * hierarchy. The reference's one-to-all / all-to-one drivers call multiple_solver once per focal point
 * (src/raster/onetoall.jl:106-151); it deletes the rows / columns of the infinite grounds and runs a fresh
 * smoothed_aggregation for every call (src/raster/advanced.jl:282-288, 307-312). Here column c of the batch keeps
 * x = 0 at the nodes ground_idx[ground_ptr[c] .. ground_ptr[c+1]) (0-based node ids; rhs entries there are ignored),
 * which is the same reduced system, preconditioned by the hierarchy of the matrix the handle was set up with.
 *   rhs, x_out:  host column-major n x nrhs arrays of the handle's value type (x_out = 0 at the grounded nodes)
 *   curr_out:    optional (may be NULL) n x nrhs node currents of each solution, computed like
 *                csgpu_solve_pairs_currents on the handle's matrix (a grounded node reports the current it sinks)
 * The residual check of core.jl:640 is evaluated on the rows of the reduced system.
 * Two things make the shared hierarchy fit these systems (csrc/pcg.h): (1) the stopping rule is CSGPU_CRIT_BOTH when
 * KRYLOV is configured -- sqrt(r0'M^-1 r0) of the UNGROUNDED hierarchy is dominated by the constant mode (gain 1 / shift)
 * whenever the right-hand side has a non-zero mean, so the reference's relative rule alone would stop at once; (2) the
 * coarsest solve is pinv-without-the-near-kernel-pairs plus the exact Galerkin answer along the candidate v_k of every
 * connected component k, v_k (v_k'b) / G_k with G_k = total conductance between column c's ground set and the free nodes of
 * the component, found per batch by sending the sets' penalty vector down the V-cycle once (26 -> 17 iterations per column
 * with an fp64 hierarchy, 51 -> 19.5 with an fp32 one: 300^2 raster, 8 one-to-all columns; two components of a 240^2 raster:
 * 36 -> 21; pair solves on the same handles: 10). Up to 256 components of the coarsest graph; more: plain pinv. */
int csgpu_solve_grounded(csgpu_handle* h, const void* rhs, int64_t nrhs, const int64_t* ground_ptr,
                         const int64_t* ground_idx, void* x_out, void* curr_out, csgpu_stats* stats);

/* csgpu_solve_grounded for right-hand sides with a few entries each -- the one-to-all / all-to-one drivers of the reference
 * (src/raster/onetoall.jl:77-158, network advanced mode src/network/advanced.jl:1-51 -> src/raster/advanced.jl:274-312): a
 * one-to-all column is ONE +1 (a dense n x nrhs upload would move 8 n bytes of zeros per column: 640 MB for 16 columns at
 * n = 5e6), and what the drivers keep of a solve is the voltage of one node and the accumulated current maps, not the n x nrhs
 * voltages. Column c: right-hand side = sum of source_val[e] at node source_idx[e], e in [source_ptr[c], source_ptr[c+1])
 * (source_val NULL: every entry is 1; several entries at one node are summed; 0-based node ids), x = 0 on column c's ground
 * set as in csgpu_solve_grounded.
 *   check_out[c]        = x[check_node[c]] of column c (`res[i] = v[1]`, onetoall.jl:141); check_node[c] < 0: 0. Both NULL or
 *                         both given.
 *   x_out, curr_out     = n x nrhs voltages / node currents as in csgpu_solve_grounded                     may be NULL
 *   cum_curr_inout[i]  += sum_c curr_c[i];  max_curr_inout[i] = max(max_curr_inout[i], max_c curr_c[i])    may be NULL
 *                         (the serial merge after the fan-out over the focal points, onetoall.jl:153-158; accumulated on the
 *                         device, one n-vector back per call)
 * Host arrays of the handle's value type. stats->device_ms is the HIP-event time of the PCG loops. */
int csgpu_solve_sources(csgpu_handle* h, int64_t nrhs, const int64_t* source_ptr, const int64_t* source_idx,
                        const void* source_val, const int64_t* ground_ptr, const int64_t* ground_idx,
                        const int64_t* check_node, void* check_out, void* x_out, void* curr_out, void* cum_curr_inout,
                        void* max_curr_inout, csgpu_stats* stats);

/* Scope row N2 / missing item "focal regions" -- effective resistance between SHORT-CIRCUITED NODE SETS on one
 * hierarchy. With focal regions (several cells per focal id) the reference merges the two regions of every pair into
 * one node each and builds a fresh graph and a fresh hierarchy per pair (_pt_file_polygons_path,
 * src/raster/pairwise.jl:72-135; create_new_polymap :369-442). A merged set is an equipotential, so on the graph in
 * which the sets are NOT merged
 *     R(I, J) = 1 / v'Av,   v = 1 on I, 0 on J,  (A v)_f = 0 on every other node,
 * one masked solve per pair (rows / columns of I u J, as in csgpu_solve_grounded), in batches of opts.batch pairs.
 *   set_ptr[nsets + 1], set_nodes:  the sets as lists of 0-based node ids
 *   src_set, dst_set [npairs]:      the two sets of every pair (indices into set_ptr)
 *   resistances [npairs] (double):  R (-1 if the energy comes out non-positive)
 * The caller filters what the reference's bookkeeping filters: sets sharing a node (R = 0), pairs whose sets share no
 * component (-1; csgpu_components), and the nodes of a set that lie in components the other set does not reach (they
 * carry no current in the merged graph either, but held at potential 1 on a regularised matrix they would add a
 * spurious eps-sized term to the energy). solver.py::focal_regions_pairwise_on_device is that caller. */
int csgpu_solve_region_pairs(csgpu_handle* h, const int64_t* set_ptr, const int64_t* set_nodes, int64_t nsets,
                             const int64_t* src_set, const int64_t* dst_set, int64_t npairs, double* resistances,
                             csgpu_stats* stats);

/* Time `reps` launches of the fine-level CSR SpMV (batch width k in {1,2,4,8,16}) with HIP events on the
 * library's stream (k up to 32); returns the average milliseconds per launch. Used by bench.py for the roofline line. */
int csgpu_spmv_bench(csgpu_handle* h, int k, int reps, double* avg_ms);

/* y = A x on the device for host vectors (tests: parity of the SpMV kernel itself). */
int csgpu_spmv_host(csgpu_handle* h, const void* x, void* y, int k);

/* y = (level `lvl` operator `which`, numbering as in csgpu_get_level_matrix) x, launched the way the V-cycle launches
 * that operator (tests: parity of the restriction / [S Q] kernels). Host arrays in the hierarchy's precision
 * (precond_bytes, else val_bytes), interleaved [ncols][k] -> [nrows][k]. For which == 5 `dots` (k doubles, may be
 * NULL) receives the fused dot products sum_i x[i][c] * y[i][c] over the first nrows entries of x.
 * which == 6 (polygon handles on the lattice path only, lvl ignored): y = Pi x, the average over every polygon's cells, for a
 * cell-space vector x [nrows * ncols of the raster][k] (column-major cell ids), and dots[c] = ||Pi x||^2 in NODE space -- the
 * norm of the MERGED system the reference checks (src/core.jl:640-641): a polygon of s cells at the value rho counts
 * (s rho)^2 -- evaluated by the kernels the PCG loop uses for its residual norms (csrc/poly.h). */
int csgpu_level_spmv_host(csgpu_handle* h, int lvl, int which, const void* x, void* y, int k, double* dots);

/* Copy level `lvl`'s operator (which: 0 = A, 1 = P, 2 = R, 3 = Q, 4 = Q^T, 5 = [S Q]; the
 * last two exist on level 0 of a two-product hierarchy only) back to the host for inspection by tests.
 * Pass NULL arrays to query sizes only. */
int csgpu_get_level_matrix(const csgpu_handle* h, int lvl, int which, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                           int32_t* rowptr, int32_t* colidx, void* vals);

/* Test hook for the lattice-form CG product (csrc/stencil.h): p_out = z + beta .* p_in (per column), y = A p_out,
 * dots[c] = p_out[:,c]' y[:,c], evaluated by the fused kernel the PCG loop uses. z, p_in, p_out: host arrays
 * [n][k] in the search direction's precision (precond_bytes, else val_bytes); y: [n][k] in val_bytes precision; beta,
 * dots: k doubles. Status 4 when the handle's matrix has no lattice form (or k == 1). */
int csgpu_dia_product_host(csgpu_handle* h, const void* z, const void* p_in, const double* beta, void* p_out, void* y,
                           int k, double* dots);

/* ---- several GPUs of one node behind ONE handle -------------------------------------------------------------------
 * The reference runs one task per source point and merges the per-task results serially (src/core.jl:262-285). Here the
 * independent unit is a chunk of pairs: a csgpu_multi owns one csgpu_handle per device (matrix + hierarchy replicated,
 * built concurrently by one host thread per device), csgpu_multi_solve_pairs deals chunks of at most opts.batch pairs
 * to the devices from a shared queue (dynamic balance; the chunk shrinks to ceil(npairs / ndevices) when there are
 * fewer batches than devices, so every GPU is busy) and every device thread writes its results straight into the
 * caller's arrays. No device-to-device traffic is needed on this path: the results are a few bytes per pair and the
 * single host process already owns them. devices == NULL: the first ndevices visible devices (ndevices <= 0: all).
 * opts->device is ignored. Semantics of the outputs and of the status code as in csgpu_solve_pairs; stats are merged
 * (sums / maxima over the chunks; solve_ms = wall time of the call; device_ms = the busiest device). */
typedef struct csgpu_multi csgpu_multi;
int csgpu_multi_setup(const void* rowptr, const void* colidx, const void* vals, int64_t n, int64_t nnz, int idx_bytes,
                      int val_bytes, int index_base, const csgpu_opts* opts, const int32_t* devices, int ndevices,
                      csgpu_multi** out);
int csgpu_multi_raster_setup(const void* cond, int64_t nrows, int64_t ncols, int val_bytes, int four_neighbors,
                             int avg_resistances, int reg, const csgpu_opts* opts, const int32_t* devices, int ndevices,
                             csgpu_multi** out);
int csgpu_multi_solve_pairs(csgpu_multi* m, const int64_t* src, const int64_t* dst, int64_t npairs,
                            const int64_t* gather_idx, int64_t ngather, void* gathered_out, void* resist_out,
                            csgpu_stats* stats);
/* The same with the reference's cumulative / maximum current maps (write_cum_maps / accum_currents!, src/out.jl:96-107,
 * merged serially after the task fan-out in src/core.jl:262-285): batches are dealt round-robin to the devices and every
 * device runs its pairs as ONE csgpu_solve_pairs_currents call, so its cumulative / maximum node-current vectors stay in its
 * HBM for the whole job; the ndevices n-vectors are combined on the host in slot order (sum / max: deterministic -- this
 * is the one place where the path moves n-sized data between devices: 0.8 GB per vector at n = 1e8, once per job).
 *   cum_curr_inout[i] += sum_p weights[p] * curr_p[i]   (weights NULL => 1)    may be NULL
 *   max_curr_inout[i]  = max(max_curr_inout[i], max_p curr_p[i])               may be NULL
 *   resist_out[p] as in csgpu_solve_pairs (may be NULL). Host arrays of the handle's value type, n = csgpu_get_info().n. */
int csgpu_multi_solve_pairs_currents(csgpu_multi* m, const int64_t* src, const int64_t* dst, int64_t npairs,
                                     const int32_t* weights, void* cum_curr_inout, void* max_curr_inout, void* resist_out,
                                     csgpu_stats* stats);
/* BASELINE configs[4] across the GPUs of a node -- the fan-out over the focal points of the one-to-all / all-to-one drivers
 * (`Threads.@spawn(f(x))` per focal point, src/raster/onetoall.jl:146-151, each f a multiple_solver call,
 * src/raster/advanced.jl:274-312) as one host thread per GPU: device slot i takes the contiguous range of columns
 * [i*nrhs/nd ...) (sizes differ by at most one column) and runs it as ONE csgpu_solve_grounded / csgpu_solve_sources call on
 * its replica of the hierarchy, writing straight into the caller's arrays; cumulative / maximum current vectors stay in each
 * device's HBM for the whole job and are combined on the host in slot order (sum / max; the merge of onetoall.jl:153-158).
 * "Replicas only across sources" (SURVEY.md section 8e): no device-to-device traffic. Arguments, outputs and status as in the
 * single-handle calls; stats merged as in csgpu_multi_solve_pairs (device_ms = the busiest device);
 * csgpu_multi_last_busy reports seconds and columns per slot. */
int csgpu_multi_solve_grounded(csgpu_multi* m, const void* rhs, int64_t nrhs, const int64_t* ground_ptr,
                               const int64_t* ground_idx, void* x_out, void* curr_out, csgpu_stats* stats);
int csgpu_multi_solve_sources(csgpu_multi* m, int64_t nrhs, const int64_t* source_ptr, const int64_t* source_idx,
                              const void* source_val, const int64_t* ground_ptr, const int64_t* ground_idx,
                              const int64_t* check_node, void* check_out, void* x_out, void* curr_out,
                              void* cum_curr_inout, void* max_curr_inout, csgpu_stats* stats);
/* number of devices of the set; handle of device slot i (for csgpu_get_info etc.; owned by the set); per-slot wall
 * seconds spent inside the last csgpu_multi_solve_pairs (busy_s: ndevices doubles, may be NULL) */
int csgpu_multi_device_count(const csgpu_multi* m);
csgpu_handle* csgpu_multi_handle(csgpu_multi* m, int slot);
int csgpu_multi_last_busy(const csgpu_multi* m, double* busy_s, int64_t* pairs_done);
void csgpu_multi_free(csgpu_multi* m);

void csgpu_free(csgpu_handle* h);
/* Device blocks released by freed handles (and by temporaries of the setup) are kept in a per-device pool and reused
 * on an exact size match -- freeing and re-allocating tens of GB through the driver costs seconds, and the reference
 * factorises again and again (per component, per focal region, per one-to-all source). This returns the pooled blocks
 * of `device` (-1: every device) to the driver and reports the bytes released. CSGPU_NO_POOL=1 disables pooling. */
int64_t csgpu_trim_memory(int device);
const char* csgpu_last_error(void);
const char* csgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CSGPU_H */
