# CircuitscapeHIPExt.jl -- reference-side binding of libcsgpu.so (NOT executed in the build image: no Julia there).
#
# Pattern: ext/CircuitscapePardisoExt.jl / ext/CircuitscapeAppleAccelerateExt.jl of Circuitscape.jl v5.17.1.
# Because libcsgpu is a plain shared library (not a Julia package) this file is `include`d from
# src/Circuitscape.jl after core.jl; together with the ~15-line patch in INTEGRATION.md it adds the solver
# alias `solver = hip` to the INI surface. Every ccall below maps 1:1 onto include/csgpu.h.
#
# Set ENV["CSGPU_LIB"] to the path of libcsgpu.so (built with hipcc --offload-arch=gfx950).

const LIBCSGPU = get(ENV, "CSGPU_LIB", "libcsgpu.so")

struct HIPAMGSolver <: Solver
    bs::Int
end

# mirror of csgpu_opts (include/csgpu.h); filled by csgpu_default_opts
mutable struct CsgpuOpts
    struct_size::Int32; device::Int32; max_levels::Int32; max_coarse::Int32; aggregation::Int32
    nu_pre::Int32; nu_post::Int32; criterion::Int32; itmax::Int32; batch::Int32; check_every::Int32; nu_coarse::Int32
    theta::Float64; omega_p::Float64; omega_s::Float64; rtol::Float64; atol::Float64
    node_row::Ptr{Int32}; node_col::Ptr{Int32}
    precond_bytes::Int32; use_graph::Int32; two_product::Int32; stencil::Int32
    explicit_check::Int32; reserved3::Int32
    # round 6: the decisions that used to be environment switches (include/csgpu.h, 0 = the default in every field)
    last_level_sweeps::Int32; enrich::Int32; enrich_steps::Int32; dia25_min_rows::Int32; dia25_prefetch::Int32
    dia25_waves::Int32; dia25_fused_j0::Int32; stream::Int32; tail_rows::Int32; poly_lattice::Int32; cellspace::Int32
    cellspace_from_csr::Int32; lattice_level1::Int32; lattice_level1_min_rows::Int32; lattice_setup::Int32; lattice_s::Int32
    lattice_q::Int32; direct_tiles::Int32; tile_pieces::Int32; direct_at::Int32; dirichlet_coarse::Int32; deflation::Int32
    tail_projection::Int32; coarse_smoother::Int32; nu_l1::Int32; nu_deep::Int32; wide_csr::Int32; fixed_k::Int32
    recompute_ap::Int32; longrow::Int32; narrow_tile::Int32; spmv_grid_cap::Int32; dia_seg::Int32; restrict_seg::Int32
    collapse_min::Int32; verbose::Int32; expander_probe::Int32; fused_restrict::Int32; sparse_init::Int32; fused_level1::Int32
    stream_min::Int64; host_stream_block::Int64
    enrich_tau::Float64; hetero_fp64_frac::Float64; poly_strength::Float64; poly_coef::Float64; poly_smin::Float64
    poly_smax::Float64; cellspace_min_frac::Float64; tile_theta::Float64; tile_split_min::Float64
    CsgpuOpts() = new()
end

mutable struct CsgpuStats
    nrhs::Int32; max_iters::Int32; total_iters::Int64; max_relres::Float64; solve_ms::Float64; device_ms::Float64
    cg_spmv_ms::Float64; cg_spmv_calls::Int64; batch::Int32; not_converged::Int32; graph_launches::Int64; polished_batches::Int64
    cg_spmv_bytes::Int64; stream_slots::Int64
    resid_ms::Float64; resid_calls::Int64; resid_bytes::Int64; resid_fused::Int32; reserved_stats::Int32
    CsgpuStats() = new(0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0)
end

mutable struct HIPFactor          # cf. PardisoFactorize (Pardiso ext :8-13): owns the device-resident hierarchy
    ptr::Ptr{Cvoid}
    function HIPFactor(ptr)
        f = new(ptr)
        finalizer(x -> (x.ptr != C_NULL && ccall((:csgpu_free, LIBCSGPU), Cvoid, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), f)
        f
    end
end

csgpu_error() = unsafe_string(ccall((:csgpu_last_error, LIBCSGPU), Cstring, ()))

function default_opts(bs::Int)
    o = CsgpuOpts()
    ccall((:csgpu_default_opts, LIBCSGPU), Cvoid, (Ref{CsgpuOpts},), o)
    o.batch = Int32(clamp(nextpow(2, bs), 1, 32))   # up to 32 columns per pass (csgpu_opts.batch)
    o
end

"""
construct_cholesky_factor(matrix, ::HIPAMGSolver) -- same hook the Pardiso / Accelerate extensions implement
(src/core.jl:519-523). SparseMatrixCSC of a symmetric Laplacian == CSR; Int64 / 1-based arrays are converted on
the device. `coords` (optional) = (rows, cols)::Tuple{Vector{Int32},Vector{Int32}} of each node's first cell.
"""
function construct_cholesky_factor(matrix::SparseMatrixCSC{T,V}, s::HIPAMGSolver; coords = nothing) where {T,V}
    o = default_opts(s.bs)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(matrix, 1)
    rc = GC.@preserve matrix coords begin
        if coords !== nothing
            o.node_row = pointer(coords[1]); o.node_col = pointer(coords[2])
        end
        ccall((:csgpu_setup, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Cint, Cint, Ref{CsgpuOpts}, Ref{Ptr{Cvoid}}),
              matrix.colptr, matrix.rowval, matrix.nzval, n, nnz(matrix), sizeof(V), sizeof(T), 1, o, h)
    end
    rc == 0 || error("csgpu_setup failed: $(csgpu_error())")
    HIPFactor(h[])
end

"solve_linear_system(factor, matrix, rhs) -- batched multi-RHS flavour (src/core.jl:646-653)."
function solve_linear_system(factor::HIPFactor, matrix::SparseMatrixCSC{T,V}, rhs::VecOrMat{T}) where {T,V}
    lhs = similar(rhs)
    st = CsgpuStats()
    rc = GC.@preserve rhs lhs ccall((:csgpu_solve_rhs, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{CsgpuStats}), factor.ptr, rhs, size(rhs, 2), lhs, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_rhs failed: $(csgpu_error())")
    lhs
end

function multiple_solve(s::HIPAMGSolver, matrix::SparseMatrixCSC{T,V}, sources::Vector{T}) where {T,V}
    factor = construct_cholesky_factor(matrix, s)
    volt = solve_linear_system(factor, matrix, sources)
    finalize(factor)
    volt
end

"""
Pair batch on the device: resistances (core.jl:232), focal voltages for the shortcut (core.jl:685-703) and, when
maps are written, full grounded voltage vectors (core.jl:231). `src`, `dst`, `gather` are 0-based local node ids.
"""
function solve_pairs(factor::HIPFactor, ::Type{T}, n::Int, src::Vector{Int64}, dst::Vector{Int64};
                     gather::Vector{Int64} = Int64[], want_voltages::Bool = false) where {T}
    np = length(src)
    res = Vector{T}(undef, np)
    gat = Matrix{T}(undef, length(gather), np)
    volt = want_voltages ? Matrix{T}(undef, n, np) : Matrix{T}(undef, 0, 0)
    st = CsgpuStats()
    rc = GC.@preserve src dst gather res gat volt ccall((:csgpu_solve_pairs, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, src, dst, np, want_voltages ? pointer(volt) : C_NULL, gather, length(gather),
              isempty(gather) ? C_NULL : pointer(gat), res, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_pairs failed: $(csgpu_error())")
    res, gat, volt, st
end

"""
Pair batch with the reference's current post-processing on the device (postprocess -> write_cur_maps, core.jl:655-683,
out.jl:46-115,178-290): node currents per pair, cumulative (`cum`) and maximum (`mx`) node currents accumulated over the
pairs (weights = number of id combinations of a node pair), optional branch currents for network mode.
"""
function solve_pairs_currents(factor::HIPFactor, ::Type{T}, n::Int, nnzA::Int, src::Vector{Int64}, dst::Vector{Int64};
                              weights::Vector{Int32} = Int32[], want_voltages = false, want_currents = true,
                              cum::Vector{T} = T[], mx::Vector{T} = T[], want_branch = false) where {T}
    np = length(src)
    res = Vector{T}(undef, np)
    volt = want_voltages ? Matrix{T}(undef, n, np) : Matrix{T}(undef, 0, 0)
    curr = want_currents ? Matrix{T}(undef, n, np) : Matrix{T}(undef, 0, 0)
    br = want_branch ? Matrix{T}(undef, nnzA, np) : Matrix{T}(undef, 0, 0)
    st = CsgpuStats()
    p(x) = isempty(x) ? C_NULL : pointer(x)
    rc = GC.@preserve src dst weights res volt curr cum mx br ccall((:csgpu_solve_pairs_currents, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
               Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, src, dst, np, p(weights), p(volt), p(curr), p(cum), p(mx), p(br), res, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_pairs_currents failed: $(csgpu_error())")
    res, volt, curr, br, st
end

"""
Device graph layer (scope row N4) for a raster without polygons: `raster_factor(cellmap, s; four_neighbors, avg_res)`
numbers the cells with conductance > 0 (construct_node_map), writes the Laplacian in HBM (construct_graph,
laplacian!, regularisation of core.jl:161) and sets up AMG -- no COO / SparseMatrixCSC on the host. `raster_nodemap`
returns the node map (1-based, 0 = NODATA), `components` the connected components (0-based dense labels).
`cellmap` is Julia's column-major Matrix; the C side wants row-major, hence the permutedims.
"""
function raster_factor(cellmap::Matrix{T}, s::HIPAMGSolver; four_neighbors = false, avg_res = false, reg = true) where {T}
    o = default_opts(s.bs)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rm = permutedims(cellmap)                      # row-major [nrows][ncols] as seen from C
    rc = GC.@preserve rm ccall((:csgpu_raster_setup, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Int64, Int64, Cint, Cint, Cint, Cint, Ref{CsgpuOpts}, Ref{Ptr{Cvoid}}),
              rm, size(cellmap, 1), size(cellmap, 2), sizeof(T), four_neighbors, avg_res, reg ? 1 : 0, o, h)
    rc == 0 || error("csgpu_raster_setup failed: $(csgpu_error())")
    HIPFactor(h[])
end

"""
compute_omniscape_current on the device (utils.jl:145-257), rasters in and out: `grounded_raster_factor(cond, ground, s)`
builds the graph with the finite ground conductances on the diagonal, `solve_raster(factor, source)` returns the
node-current map. Many moving windows can be stacked into one raster separated by NODATA rows: one setup, one PCG.
"""
function grounded_raster_factor(cellmap::Matrix{T}, ground::Matrix{T}, s::HIPAMGSolver; four_neighbors = false) where {T}
    o = default_opts(1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rm = permutedims(cellmap); gm = permutedims(ground)
    rc = GC.@preserve rm gm ccall((:csgpu_raster_setup_grounded, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Cint, Cint, Cint, Ref{CsgpuOpts}, Ref{Ptr{Cvoid}}),
              rm, gm, size(cellmap, 1), size(cellmap, 2), sizeof(T), four_neighbors, 0, 0, o, h)
    rc == 0 || error("csgpu_raster_setup_grounded failed: $(csgpu_error())")
    HIPFactor(h[])
end

function solve_raster(factor::HIPFactor, source::Matrix{T}) where {T}
    sm = permutedims(source)
    cur = similar(sm)
    st = CsgpuStats()
    rc = GC.@preserve sm cur ccall((:csgpu_solve_raster, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}), factor.ptr, sm, cur, C_NULL, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_raster failed: $(csgpu_error())")
    permutedims(cur)
end

function raster_nodemap(factor::HIPFactor)
    r = Ref{Int64}(0); c = Ref{Int64}(0)
    ccall((:csgpu_raster_nodemap, LIBCSGPU), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ref{Int64}, Ref{Int64}), factor.ptr, C_NULL, r, c)
    nm = Matrix{Int32}(undef, c[], r[])            # row-major from C == transposed column-major
    rc = GC.@preserve nm ccall((:csgpu_raster_nodemap, LIBCSGPU), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int64}, Ptr{Int64}),
                               factor.ptr, nm, C_NULL, C_NULL)
    rc == 0 || error("csgpu_raster_nodemap failed: $(csgpu_error())")
    permutedims(nm)
end

function components(factor::HIPFactor, n::Int)
    lab = Vector{Int32}(undef, n); nc = Ref{Int64}(0)
    rc = GC.@preserve lab ccall((:csgpu_components, LIBCSGPU), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ref{Int64}), factor.ptr, lab, nc)
    rc == 0 || error("csgpu_components failed: $(csgpu_error())")
    lab, Int(nc[])
end

"""
One-to-all / all-to-one on ONE hierarchy (scope row N2): column c of `rhs` is solved with the nodes `grounds[c]`
(0-based) tied directly to ground -- what `multiple_solver` does per focal point by deleting rows / columns and
factorising again (raster/onetoall.jl:106-151, raster/advanced.jl:282-312). Returns voltages (n x nrhs) and, optionally,
node currents.
"""
function solve_grounded(factor::HIPFactor, rhs::Matrix{T}, grounds::Vector{Vector{Int64}}; want_currents = false) where {T}
    n, nrhs = size(rhs)
    gptr = Int64[0; cumsum(length.(grounds))]
    gidx = isempty(grounds) ? Int64[] : reduce(vcat, grounds)
    isempty(gidx) && (gidx = Int64[0])
    x = similar(rhs)
    cur = want_currents ? similar(rhs) : Matrix{T}(undef, 0, 0)
    st = CsgpuStats()
    rc = GC.@preserve rhs gptr gidx x cur ccall((:csgpu_solve_grounded, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, rhs, nrhs, gptr, gidx, x, want_currents ? pointer(cur) : C_NULL, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_grounded failed: $(csgpu_error())")
    x, cur, st
end

"""
csgpu_solve_sources / csgpu_multi_solve_sources: Dirichlet-masked solves whose right-hand sides have a few entries each -- the
columns the one-to-all / all-to-one drivers build (raster/onetoall.jl:106-117: one +1 per one-to-all column) -- handed over
as lists, with what those drivers keep of a solve: the voltage of one node per column (`res[i] = v[1]`, onetoall.jl:141) and
the cumulative / maximum node-current vectors (onetoall.jl:153-158), accumulated on the device. No n x nrhs array crosses
PCIe unless `want_voltages` / `want_currents` ask for it. `sources[c]` / `grounds[c]`: 0-based node ids; `values[c]`: the
entries (empty = ones); `check[c]`: node whose voltage is returned (-1: none). With a `HIPMultiFactor` the columns are dealt
over the GPUs of the node (the fan-out of onetoall.jl:146-151 as one host thread per GPU).
"""
function solve_sources(factor::Union{HIPFactor,HIPMultiFactor}, ::Type{T}, n::Int, sources::Vector{Vector{Int64}},
                       grounds::Vector{Vector{Int64}}; values::Vector{Vector{T}} = Vector{T}[], check::Vector{Int64} = Int64[],
                       want_voltages = false, want_currents = false, cum::Vector{T} = T[], mx::Vector{T} = T[]) where {T}
    nrhs = length(sources)
    sptr = Int64[0; cumsum(length.(sources))]
    sidx = isempty(sources) ? Int64[] : reduce(vcat, sources)
    isempty(sidx) && (sidx = Int64[0])
    sval = isempty(values) ? T[] : reduce(vcat, values)
    gptr = Int64[0; cumsum(length.(grounds))]
    gidx = isempty(grounds) ? Int64[] : reduce(vcat, grounds)
    isempty(gidx) && (gidx = Int64[0])
    cout = Vector{T}(undef, length(check))
    volt = want_voltages ? Matrix{T}(undef, n, nrhs) : Matrix{T}(undef, 0, 0)
    cur = want_currents ? Matrix{T}(undef, n, nrhs) : Matrix{T}(undef, 0, 0)
    st = CsgpuStats()
    rc = GC.@preserve sptr sidx sval gptr gidx check cout volt cur cum mx begin
        pval = isempty(sval) ? C_NULL : pointer(sval)
        pchk = isempty(check) ? C_NULL : pointer(check)
        pout = isempty(check) ? C_NULL : pointer(cout)
        pv = want_voltages ? pointer(volt) : C_NULL
        pc = want_currents ? pointer(cur) : C_NULL
        pcum = isempty(cum) ? C_NULL : pointer(cum)
        pmx = isempty(mx) ? C_NULL : pointer(mx)
        if factor isa HIPMultiFactor
            ccall((:csgpu_multi_solve_sources, LIBCSGPU), Cint,
                  (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid},
                   Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
                  factor.ptr, nrhs, sptr, sidx, pval, gptr, gidx, pchk, pout, pv, pc, pcum, pmx, st)
        else
            ccall((:csgpu_solve_sources, LIBCSGPU), Cint,
                  (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid},
                   Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
                  factor.ptr, nrhs, sptr, sidx, pval, gptr, gidx, pchk, pout, pv, pc, pcum, pmx, st)
        end
    end
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_sources failed: $(csgpu_error())")
    cout, volt, cur, st
end

"csgpu_multi_solve_grounded: `solve_grounded` with the columns dealt over the GPUs of the node (dense right-hand sides)."
function solve_grounded(factor::HIPMultiFactor, rhs::Matrix{T}, grounds::Vector{Vector{Int64}}; want_currents = false) where {T}
    n, nrhs = size(rhs)
    gptr = Int64[0; cumsum(length.(grounds))]
    gidx = isempty(grounds) ? Int64[] : reduce(vcat, grounds)
    isempty(gidx) && (gidx = Int64[0])
    x = similar(rhs)
    cur = want_currents ? similar(rhs) : Matrix{T}(undef, 0, 0)
    st = CsgpuStats()
    rc = GC.@preserve rhs gptr gidx x cur ccall((:csgpu_multi_solve_grounded, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, rhs, nrhs, gptr, gidx, x, want_currents ? pointer(cur) : C_NULL, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_multi_solve_grounded failed: $(csgpu_error())")
    x, cur, st
end

"""
One-to-all / all-to-one with ONE graph build and ONE AMG setup for all focal points (scope rows N2 + N4; the Julia twin of
`solver.py::onetoall_on_device`). `onetoall_kernel` (raster/onetoall.jl:13-162) builds the graph once but calls
`advanced_kernel` -> `multiple_solver` per focal point, which deletes the grounded rows / columns and factorises again
(raster/advanced.jl:282-312). Here every focal point is one column of `csgpu_solve_grounded` -- the same reduced systems on
the hierarchy of the ungrounded Laplacian, `s.bs` points per PCG -- and the node currents come from the device
(out.jl:178-207). Applies to rasters without polygons, single-cell focal points, no variable strengths, no included pairs
(`onetoall_on_device_applies`); everything else keeps the reference's driver, which reaches the device through
`multiple_solve(::HIPAMGSolver, ...)`. Dispatch (INTEGRATION.md section 2): first line of `onetoall_kernel`,
`s = get_solver(cfg); s isa HIPAMGSolver && onetoall_on_device_applies(data) && return onetoall_on_device(data, flags, cfg, s)`.
"""
onetoall_on_device_applies(data) =
    isempty(data.strengths) && isempty(data.included_pairs) && isempty(data.polymap) &&
    length(unique(data.points_rc[3])) == length(data.points_rc[3])

function onetoall_on_device(data, flags, cfg, s::HIPAMGSolver)
    gmap = data.cellmap
    T = eltype(gmap)
    hbmeta = data.hbmeta
    rows, cols, ids = data.points_rc                  # 1-based cells and point ids (raster/onetoall.jl:17)
    np = length(ids)
    of = flags.outputflags
    one_to_all = flags.is_onetoall
    want_cur = of.write_cur_maps || of.write_cum_cur_map_only
    res = fill(T(-1), np)
    cum = initialize_cum_maps(gmap, of.write_max_cur_maps)
    np == 1 && return hcat(ids, res)                  # (a single point: sum(point_map) == n, onetoall.jl:102-105)
    # no regularisation shift: the grounded systems are non-singular (advanced.jl never adds one)
    factor = raster_factor(gmap, s; four_neighbors = flags.four_neighbors, avg_res = flags.avg_res, reg = false)
    nodemap = raster_nodemap(factor)
    n = Int(maximum(nodemap))
    node = Int64[nodemap[rows[i], cols[i]] - 1 for i in 1:np]          # 0-based; -1: the focal cell is NODATA
    comp, _ = components(factor, n)
    sources = Vector{Vector{Int64}}(undef, np)        # the columns as the driver builds them: a few unit entries each
    grounds = Vector{Vector{Int64}}(undef, np)
    solvable = falses(np)
    gcomps = Vector{Set{Int32}}(undef, np)
    for i in 1:np
        others = Int64[node[k] for k in 1:np if k != i && node[k] >= 0]
        own = node[i] >= 0 ? Int64[node[i]] : Int64[]
        src, gnd = one_to_all ? (own, others) : (others, own)
        gcomps[i] = Set{Int32}(comp[g + 1] for g in gnd)
        # a component without a ground is not solved (advanced.jl:186-191)
        sources[i] = Int64[q for q in src if comp[q + 1] in gcomps[i]]
        solvable[i] = !isempty(sources[i])
        grounds[i] = gnd
    end
    if !of.write_volt_maps && !(of.write_cur_maps && !of.write_cum_cur_map_only)
        # Nothing per focal point is written: the driver keeps the voltage of the source (`res[i] = v[1]`, onetoall.jl:141)
        # and the accumulated current maps (onetoall.jl:153-158) -- csgpu_solve_sources hands back exactly those (sparse
        # right-hand sides in, np voltages + at most two n-vectors out; the maps are accumulated on the device).
        node_cum = want_cur ? zeros(T, n) : T[]
        node_max = (want_cur && of.write_max_cur_maps) ? zeros(T, n) : T[]
        vchk, _, _, _ = solve_sources(factor, T, n, sources, grounds; check = Int64[one_to_all ? node[i] : -1 for i in 1:np],
                                      cum = node_cum, mx = node_max)
        finalize(factor)
        for i in 1:np
            res[i] = one_to_all ? ((solvable[i] && vchk[i] != 0) ? vchk[i] : T(-1)) : (solvable[i] ? T(0) : T(-1))
        end
        if want_cur
            for k in eachindex(nodemap)
                nodemap[k] == 0 && continue
                cum.cum_curr[k] += node_cum[nodemap[k]]
                of.write_max_cur_maps && (cum.max_curr[k] = max(cum.max_curr[k], node_max[nodemap[k]]))
            end
            write_cum_maps(cum, gmap, cfg, hbmeta, of.write_max_cur_maps, of.write_cum_cur_map_only)
        end
        return hcat(ids, res)
    end
    rhs = zeros(T, n, np)
    for i in 1:np, q in sources[i]
        rhs[q + 1, i] = one(T)
    end
    volt, curr, _ = solve_grounded(factor, rhs, grounds; want_currents = want_cur)
    finalize(factor)                                  # (releases the device-resident hierarchy now; HIPFactor finalizer)
    for i in 1:np
        # a component without a ground is not part of the column's system; the reference has no voltage there
        outside = Bool[!(comp[k] in gcomps[i]) for k in 1:n]
        volt[outside, i] .= 0
        want_cur && (curr[outside, i] .= 0)
        vmap = zeros(T, size(gmap))
        cmap = want_cur ? zeros(T, size(gmap)) : zeros(T, 0, 0)
        for k in eachindex(nodemap)
            nodemap[k] == 0 && continue
            vmap[k] = volt[nodemap[k], i]
            want_cur && (cmap[k] = curr[nodemap[k], i])
        end
        if one_to_all
            v = vmap[rows[i], cols[i]]
            res[i] = (solvable[i] && v != 0) ? v : T(-1)               # (advanced.jl:252-262: voltage at the source / 1 A)
        else
            res[i] = solvable[i] ? T(0) : T(-1)                        # (advanced.jl:263-267)
        end
        name = "_$(ids[i])"
        of.write_volt_maps && write_grid(vmap, name, cfg, hbmeta, gmap, voltage = true)
        if want_cur
            !of.write_cum_cur_map_only && of.write_cur_maps && write_grid(cmap, name, cfg, hbmeta, gmap)
            cum.cum_curr .+= cmap                                      # (onetoall.jl:153-158)
            of.write_max_cur_maps && (cum.max_curr .= max.(cum.max_curr, cmap))
        end
    end
    want_cur && write_cum_maps(cum, gmap, cfg, hbmeta, of.write_max_cur_maps, of.write_cum_cur_map_only)
    hcat(ids, res)
end

"""
Effective resistance between short-circuited node sets on ONE hierarchy (csgpu_solve_region_pairs): what
`_pt_file_polygons_path` (raster/pairwise.jl:72-135) obtains with a fresh graph and hierarchy per pair of focal regions.
`sets` hold 1-based node ids of the graph in which the regions are NOT merged; `pairs[p] = (i, j)` indexes into `sets`.
Returns R (-1: no current flows between the two sets).
"""
function solve_region_pairs(factor::HIPFactor, sets::Vector{Vector{Int64}}, pairs::Vector{Tuple{Int,Int}})
    isempty(pairs) && return Float64[], CsgpuStats()
    sptr = Int64[0; cumsum(length.(sets))]
    snodes = isempty(sets) ? Int64[] : reduce(vcat, sets) .- 1
    a = Int64[p[1] - 1 for p in pairs]
    b = Int64[p[2] - 1 for p in pairs]
    R = zeros(Float64, length(pairs))
    st = CsgpuStats()
    rc = GC.@preserve sptr snodes a b R ccall((:csgpu_solve_region_pairs, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}, Ref{CsgpuStats}),
              factor.ptr, sptr, snodes, length(sets), a, b, length(pairs), R, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_solve_region_pairs failed: $(csgpu_error())")
    R, st
end

"""
Raster with short-circuit polygons, graph layer on the device (construct_node_map with a polymap,
raster/pairwise.jl:276-301): `polymap` > 0 names the polygon of a cell.
"""
function raster_factor(cellmap::Matrix{T}, polymap::Matrix{<:Integer}, s::HIPAMGSolver; four_neighbors = false,
                       avg_res = false) where {T}
    o = default_opts(s.bs)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rm = permutedims(cellmap); pm = Matrix{Int32}(permutedims(polymap))
    rc = GC.@preserve rm pm ccall((:csgpu_raster_setup_poly, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int32}, Int64, Int64, Cint, Cint, Cint, Cint, Ref{CsgpuOpts}, Ref{Ptr{Cvoid}}),
              rm, pm, size(cellmap, 1), size(cellmap, 2), sizeof(T), four_neighbors, avg_res, 1, o, h)
    rc == 0 || error("csgpu_raster_setup_poly failed: $(csgpu_error())")
    HIPFactor(h[])
end

"""
All GPUs of the node behind one handle (csgpu_multi_*): replaces the Threads.@spawn fan-out and the serial merge of
core.jl:262-285 by ONE call; results land in `res` / `gat` directly.
"""
mutable struct HIPMultiFactor
    ptr::Ptr{Cvoid}
    function HIPMultiFactor(ptr)
        f = new(ptr)
        finalizer(x -> (x.ptr != C_NULL && ccall((:csgpu_multi_free, LIBCSGPU), Cvoid, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), f)
        f
    end
end

# `coords` as in construct_cholesky_factor: with the raster cell of every node each device builds the cell-space lattice
# hierarchy a raster with NODATA cells gets from the single-device factor (csgpu_multi_setup forwards the options to every
# device; ADVICE r5: without them every device fell back to the general CSR hierarchy).
function construct_multi_factor(matrix::SparseMatrixCSC{T,V}, s::HIPAMGSolver; coords = nothing, ndevices::Int = 0) where {T,V}
    o = default_opts(s.bs)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve matrix coords begin
        if coords !== nothing
            o.node_row = pointer(coords[1]); o.node_col = pointer(coords[2])
        end
        ccall((:csgpu_multi_setup, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Cint, Cint, Ref{CsgpuOpts}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
              matrix.colptr, matrix.rowval, matrix.nzval, size(matrix, 1), nnz(matrix), sizeof(V), sizeof(T), 1, o,
              C_NULL, ndevices, h)
    end
    rc == 0 || error("csgpu_multi_setup failed: $(csgpu_error())")
    HIPMultiFactor(h[])
end

"""
`maps_on_all_devices(flags, s, np)`: whether a maps-on pair list goes to ALL GPUs of the node in one call
(csgpu_multi_solve_pairs_currents) instead of through the single-device factor: only cumulative / maximum maps are written
(linear accumulation, nothing per pair to bring back), more than one device is visible, and the list holds at least two
batches per device -- a replicated hierarchy per device is not worth building for less. The caller then builds the
multi-device factor INSTEAD of the single-device one (two hierarchies of one matrix never share device 0).
"""
function maps_on_all_devices(flags, s::HIPAMGSolver, np::Int)
    of = flags.outputflags
    flags.is_raster || return false
    (of.log_transform_maps || of.write_volt_maps) && return false
    (of.write_cur_maps && !of.write_cum_cur_map_only) && return false
    nd = device_count()
    nd > 1 && np >= 2 * max(1, s.bs) * nd
end

function solve_pairs(factor::HIPMultiFactor, ::Type{T}, src::Vector{Int64}, dst::Vector{Int64};
                     gather::Vector{Int64} = Int64[]) where {T}
    np = length(src)
    res = Vector{T}(undef, np)
    gat = Matrix{T}(undef, length(gather), np)
    st = CsgpuStats()
    rc = GC.@preserve src dst gather res gat ccall((:csgpu_multi_solve_pairs, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, src, dst, np, gather, length(gather), isempty(gather) ? C_NULL : pointer(gat), res, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_multi_solve_pairs failed: $(csgpu_error())")
    res, gat, st
end

"""
csgpu_multi_solve_pairs_currents: the pair list dealt over all GPUs of the node with the cumulative / maximum node-current
vectors (out.jl:96-107) accumulated per device and combined on return -- the maps-on counterpart of `solve_pairs` above for
runs that write cumulative / maximum maps only. `cum` / `mx`: length-n vectors updated in place (empty = not wanted).
"""
function solve_pairs_currents(factor::HIPMultiFactor, ::Type{T}, src::Vector{Int64}, dst::Vector{Int64};
                              weights::Vector{Int32} = Int32[], cum::Vector{T} = T[], mx::Vector{T} = T[]) where {T}
    np = length(src)
    res = Vector{T}(undef, np)
    st = CsgpuStats()
    rc = GC.@preserve src dst weights cum mx res ccall((:csgpu_multi_solve_pairs_currents, LIBCSGPU), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CsgpuStats}),
              factor.ptr, src, dst, np, isempty(weights) ? C_NULL : pointer(weights),
              isempty(cum) ? C_NULL : pointer(cum), isempty(mx) ? C_NULL : pointer(mx), res, st)
    rc == 1 && error("CG solver did not converge: relative residual $(st.max_relres) exceeds tolerance 1e-4")
    rc == 0 || error("csgpu_multi_solve_pairs_currents failed: $(csgpu_error())")
    res, st
end

device_count() = Int(ccall((:csgpu_device_count, LIBCSGPU), Cint, ()))

"""
`node_cell_table(nodemap)`: 0-based raster row / column of the FIRST cell (column-major order) of every node id of the
node map, built ONCE per problem (one pass over the raster); `node_coords(table, comp)` then picks the entries of a
connected component (any order of `comp`), or returns `nothing` -- with a warning -- when a node of the component has no
cell, in which case the component runs without the lattice / 3 x 3-tile paths.
"""
function node_cell_table(nodemap::Matrix{V}) where {V}
    nmax = isempty(nodemap) ? 0 : Int(maximum(nodemap))
    rows = fill(Int32(-1), nmax); cols = fill(Int32(-1), nmax)
    for j in axes(nodemap, 2), i in axes(nodemap, 1)
        id = Int(nodemap[i, j])
        (id == 0 || rows[id] >= 0) && continue
        rows[id] = Int32(i - 1); cols[id] = Int32(j - 1)
    end
    (rows, cols)
end

function node_coords(table::Tuple{Vector{Int32},Vector{Int32}}, comp::Vector{V}) where {V}
    rows = Vector{Int32}(undef, length(comp)); cols = Vector{Int32}(undef, length(comp))
    for (k, id) in enumerate(comp)
        if id > length(table[1]) || table[1][id] < 0
            @warn("node $id of a connected component has no raster cell: the component runs on the CSR kernels")
            return nothing
        end
        rows[k] = table[1][id]; cols[k] = table[2][id]
    end
    (rows, cols)
end

"""
Scatter a node vector into the raster through a component's local node map (the loop of `_create_current_maps`,
out.jl:163-173, and `_create_voltage_map`, out.jl:421-434).
"""
function scatter_nodes(vals::AbstractVector{T}, local_nodemap, hbmeta) where {T}
    m = zeros(T, hbmeta.nrows, hbmeta.ncols)
    for j in axes(local_nodemap, 2), i in axes(local_nodemap, 1)
        idx = local_nodemap[i, j]
        idx == 0 && continue
        m[i, j] = vals[idx]
    end
    m
end

"""
`postprocess` with maps on (core.jl:655-683 -> write_volt_maps / write_cur_maps, out.jl:29-115, 388-410) for the pair
list of ONE connected component, in chunks of `s.bs` pairs through `solve_pairs_currents`:

  * node currents (get_node_currents, out.jl:178-207, incl. the 1e-8 * max branch-current threshold) come from the device;
  * raster, linear maps: cumulative and maximum node currents are accumulated ON THE DEVICE across all chunks (weights =
    number of id combinations a node pair serves -- the reference post-processes once per combination) and scattered into
    `cum.cum_curr` / `cum.max_curr` once per component; per-pair current vectors cross PCIe only when per-pair maps are
    written (`write_cur_maps && !write_cum_cur_map_only`) or `log_transform_maps` makes the accumulation non-linear;
  * voltages cross PCIe only when `write_volt_maps` is set;
  * network mode: node and branch current tables per pair from the device's node / branch currents, accumulated into
    `cum.cum_node_curr` / `cum.cum_branch_curr` like out.jl:60-80.

Host memory is O(n * bs) (O(nnz * bs) in network mode), as in the reference's batched driver (core.jl:448-493).
Returns the resistance of every pair of `src0` / `dst0`.
"""
function solve_pairs_with_maps!(factor::Union{HIPFactor,HIPMultiFactor}, s::HIPAMGSolver, matrix::SparseMatrixCSC{T,V},
                                component_data, src0::Vector{Int64}, dst0::Vector{Int64}, fan, points, orig_pts, cum,
                                flags, cfg) where {T,V}
    of = flags.outputflags
    n = size(matrix, 1)
    np = length(src0)
    res = Vector{T}(undef, np)
    comp = component_data.cc
    hbmeta = component_data.hbmeta
    cellmap = component_data.cellmap
    local_nodemap = component_data.local_nodemap
    bs = max(1, s.bs)
    if flags.is_raster
        linear = !of.log_transform_maps
        per_pair_cur = (of.write_cur_maps && !of.write_cum_cur_map_only) || !linear
        node_cum = linear ? zeros(T, n) : T[]
        node_max = (linear && of.write_max_cur_maps) ? zeros(T, n) : T[]
        ncombos = 0
        # a multi-device factor (the caller decided: maps_on_all_devices): the whole pair list in ONE call over all devices
        # (csgpu_multi_solve_pairs_currents; the merge of core.jl:262-285 happens inside the library)
        multi = factor isa HIPMultiFactor
        if multi
            w = Int32[length(fan[p]) for p in 1:np]
            ncombos = sum(w)
            res, _ = solve_pairs_currents(factor, T, src0, dst0; weights = w, cum = node_cum, mx = node_max)
        end
        for lo in (multi ? (1:0) : (1:bs:np))
            hi = min(lo + bs - 1, np)
            w = Int32[length(fan[p]) for p in lo:hi]
            ncombos += sum(w)
            r, volt, curr, _, _ = solve_pairs_currents(factor, T, n, nnz(matrix), src0[lo:hi], dst0[lo:hi]; weights = w,
                                      want_voltages = of.write_volt_maps, want_currents = per_pair_cur,
                                      cum = node_cum, mx = node_max)
            res[lo:hi] = r
            for (k, p) in enumerate(lo:hi), (ci, cj) in fan[p]
                name = "_$(orig_pts[ci])_$(orig_pts[cj])"
                if of.write_volt_maps
                    out = Output(points, volt[:, k], (orig_pts[ci], orig_pts[cj]), (V(src0[p] + 1), V(dst0[p] + 1)), r[k], V(cj), cum)
                    write_volt_maps(name, out, component_data, flags, cfg)
                end
                if per_pair_cur
                    cmap = scatter_nodes(view(curr, :, k), local_nodemap, hbmeta)
                    process_grid!(cmap, cellmap, hbmeta, log_transform = of.log_transform_maps,
                                  set_null_to_nodata = of.set_null_currents_to_nodata)
                    if !linear                                     # log-transformed maps accumulate map by map (out.jl:96-107)
                        lock(cum.lock) do
                            cum.cum_curr .+= cmap
                            of.write_max_cur_maps && (cum.max_curr .= max.(cum.max_curr, cmap))
                        end
                    end
                    !of.write_cum_cur_map_only && of.write_cur_maps && write_grid(cmap, name, cfg, hbmeta)
                end
            end
        end
        if linear
            # one scatter per component; the NODATA value lands in the cumulative map once per id combination, exactly as
            # process_grid! + `cum_curr .+= cmap` do per pair in the reference (out.jl:89-100)
            cmap = scatter_nodes(node_cum, local_nodemap, hbmeta)
            if of.set_null_currents_to_nodata
                for i in eachindex(cmap)
                    cellmap[i] == 0 && (cmap[i] = hbmeta.nodata * ncombos)
                end
            end
            lock(cum.lock) do
                cum.cum_curr .+= cmap
                if of.write_max_cur_maps
                    mmap = scatter_nodes(node_max, local_nodemap, hbmeta)
                    process_grid!(mmap, cellmap, hbmeta, set_null_to_nodata = of.set_null_currents_to_nodata)
                    cum.max_curr .= max.(cum.max_curr, mmap)
                end
            end
        end
    else
        # network mode (out.jl:46-84): the stored entries (row < col) of the symmetric matrix, in the order
        # _get_branch_currents enumerates them (column by column of the upper triangle, out.jl:223-240)
        I = V[]; J = V[]; K = Int[]
        for i in 1:n, k in nzrange(matrix, i)
            row = matrix.rowval[k]
            i > row && (push!(I, row); push!(J, V(i)); push!(K, k))
        end
        coord_index = Dict{Tuple{Int,Int},Int}()                            # (ADVICE r4: was a findfirst scan per branch per pair)
        for (i, c) in enumerate(cum.coords)
            coord_index[(Int(c[1]), Int(c[2]))] = i
        end
        for lo in 1:bs:np
            hi = min(lo + bs - 1, np)
            r, volt, curr, br, _ = solve_pairs_currents(factor, T, n, nnz(matrix), src0[lo:hi], dst0[lo:hi];
                                      want_voltages = of.write_volt_maps, want_currents = true, want_branch = true)
            res[lo:hi] = r
            for (k, p) in enumerate(lo:hi), (ci, cj) in fan[p]
                name = "_$(orig_pts[ci])_$(orig_pts[cj])"
                of.write_volt_maps && write_voltages(cfg.output_file, name, volt[:, k], comp)
                # |g (v_i - v_j)| is symmetric in (i, j): the device stores it at the (row < col) position of its CSR
                # arrays = the mirror image of entry K of the upper triangle; read it through the transpose position
                bvals = T[max(br[kk, k], br[mirror_entry(matrix, kk), k]) for kk in K]
                branch_currents_array = _convert_to_3col(sparse(I, J, bvals, n, n), comp)
                node_currents_array = _append_name_to_node_currents(curr[:, k], comp)
                lock(cum.lock) do
                    for i in 1:size(branch_currents_array, 1)
                        a1 = (Int(branch_currents_array[i, 1]), Int(branch_currents_array[i, 2]))
                        idx = get(coord_index, a1, 0)
                        idx == 0 && (idx = coord_index[(a1[2], a1[1])])
                        cum.cum_branch_curr[idx] += branch_currents_array[i, 3]
                    end
                    for i in 1:size(node_currents_array, 1)
                        cum.cum_node_curr[Int(node_currents_array[i, 1])] += node_currents_array[i, 2]
                    end
                end
                write_currents(node_currents_array, branch_currents_array, name, cfg)
            end
        end
    end
    res
end

"position of the mirror image (column `rowval[k]`, row = the column of entry k) of stored entry `k` of a symmetric CSC matrix"
function mirror_entry(matrix::SparseMatrixCSC, k::Int)
    col = searchsortedlast(matrix.colptr, k)          # column holding entry k
    row = matrix.rowval[k]
    rng = nzrange(matrix, row)
    first(rng) + searchsortedfirst(view(matrix.rowval, rng), col) - 1
end

"""
solve(prob, ::HIPAMGSolver, flags, cfg, log) -- the pairwise kernel behind `single_ground_all_pairs` (src/core.jl:81-83)
for `solver = hip`. Same results and the same bookkeeping as solve(prob, ::AMGSolver, ...) (src/core.jl:96-305):
-1 / 0 conventions, `smash_repeats!` for ids sharing a node, exclude pairs, the resistance shortcut (anchor point only,
`update_voltmatrix!` / `update_shortcut_resistances!`), `postprocess` per id combination, `save_resistances`. What differs
is the shape of the work: per connected component the hierarchy is built ONCE on the device and the component's whole
pair list goes down in ONE `solve_pairs` call (batched `s.bs` right-hand sides per pass) instead of one
`Threads.@spawn` task per source point each cloning the AMG workspace (core.jl:173-180, 262-285).
With maps on, the pair list goes down in CHUNKS of `s.bs` pairs through `solve_pairs_currents`: node currents are
computed on the device, the cumulative and maximum node currents are accumulated there across chunks, voltages cross
PCIe only when `write_volt_maps` is set -- host memory is O(n * bs) like the reference's batched driver
(core.jl:448-493), never n x npairs.
`circuitscape.jl_amd/solver.py::solve` is the Python mirror of this method (same structure, same device calls; it is
what the parity tests run against the reference's golden files -- this file cannot be executed in the build image).
"""
function solve(prob::GraphProblem{T,V}, s::HIPAMGSolver, flags, cfg, log)::Matrix{T} where {T,V}
    a = prob.G
    cc = prob.cc
    points = prob.points
    exclude = prob.exclude_pairs
    orig_pts = prob.user_points
    of = flags.outputflags
    numpoints = size(points, 1)
    cum = prob.cum

    @info("Graph has $(size(a,1)) nodes, $numpoints focal points and $(length(cc)) connected components")
    num_pairs, _ = get_num_pairs(cc, points, exclude, orig_pts)
    log && @info("Total number of pair solves = $num_pairs")

    resistances = -1 * ones(T, numpoints, numpoints)
    voltmatrix = zeros(T, numpoints, numpoints)
    shortcut_res = deepcopy(resistances)

    use_shortcut = flags.is_raster && !of.write_volt_maps && !of.write_cur_maps && !of.write_cum_cur_map_only &&
                   !of.write_max_cur_maps && isempty(exclude)
    if use_shortcut
        @info("Triggering resistance calculation shortcut")
        num_pairs, _ = get_num_pairs_shortcut(cc, points, exclude, orig_pts)
        @info("Total number of pair solves has been reduced to $num_pairs ")
    end
    shortcut = Shortcut(use_shortcut, voltmatrix, shortcut_res)
    want_maps = !use_shortcut && (of.write_volt_maps || of.write_cur_maps || of.write_cum_cur_map_only || of.write_max_cur_maps)
    # raster cell of every node, one pass over the node map for ALL components (seeds 3 x 3 tiles / the cell-space lattice)
    cell_table = (flags.is_raster && !isempty(prob.nodemap)) ? node_cell_table(prob.nodemap) : nothing

    for comp in cc
        csub = unique(filter(x -> x in comp, points))
        isempty(csub) && continue

        matrix = a[comp, comp]
        matrix.nzval .+= eps(eltype(matrix)) * norm(matrix.nzval)            # core.jl:161
        n = size(matrix, 1)
        local_of = Dict{V,Int64}(node => Int64(findfirst(isequal(node), comp)) for node in csub)

        # ---- pair list of the component: one right-hand side per distinct (src_node, dst_node)
        src0 = Int64[]; dst0 = Int64[]                                       # 0-based local node ids for the device
        fan = Vector{Vector{Tuple{Int,Int}}}()                               # id combinations served by each solve
        nsrc = use_shortcut ? 1 : length(csub)                               # shortcut: the anchor point only (core.jl:256-260)
        for pi in 1:nsrc
            src_node = csub[pi]
            src_idx = findall(isequal(src_node), points)
            use_shortcut || smash_repeats!(resistances, src_idx)            # (the shortcut branch discards them, core.jl:259)
            for pj in pi+1:length(csub)
                dst_node = csub[pj]
                dst_idx = findall(isequal(dst_node), points)
                combos = [(ci, cj) for ci in src_idx for cj in dst_idx if (orig_pts[ci], orig_pts[cj]) ∉ exclude]
                isempty(combos) && continue
                push!(src0, local_of[src_node] - 1); push!(dst0, local_of[dst_node] - 1); push!(fan, combos)
            end
        end

        if !isempty(src0)
            # raster coordinates of the component's nodes seed the aggregation with 3 x 3 tiles (csgpu_opts.node_row/col)
            coords = cell_table === nothing ? nothing : node_coords(cell_table, comp)
            all_devices = want_maps && maps_on_all_devices(flags, s, length(src0))
            factor = @timeit CSTIMER "construct preconditioner" (all_devices ?
                         construct_multi_factor(matrix, s; coords = coords) :
                         construct_cholesky_factor(matrix, s; coords = coords))
            if want_maps
                # maps on: chunks of s.bs pairs, currents on the device (see the doc string)
                component_data = ComponentData(comp, matrix, construct_local_node_map(prob.nodemap, comp, prob.polymap),
                                               prob.hbmeta, prob.cellmap)
                res = @timeit CSTIMER "solve and accumulate pairs" solve_pairs_with_maps!(factor, s, matrix, component_data,
                                                  src0, dst0, fan, points, orig_pts, cum, flags, cfg)
                finalize(factor)
                for (p, combos) in enumerate(fan), (ci, cj) in combos
                    resistances[ci, cj] = res[p]
                    resistances[cj, ci] = res[p]
                end
            else
                focal_in_comp = findall(x -> x in comp, points)
                gather = use_shortcut ? Int64[findfirst(isequal(points[i]), comp) - 1 for i in focal_in_comp] : Int64[]
                res, gat, _, _ = @timeit CSTIMER "solve and accumulate pairs" solve_pairs(factor, T, n, src0, dst0;
                                                      gather = gather, want_voltages = false)
                finalize(factor)
                for (p, combos) in enumerate(fan)
                    r = res[p]
                    for (ci, cj) in combos
                        resistances[ci, cj] = r
                        resistances[cj, ci] = r
                        if use_shortcut
                            # update_voltmatrix! (core.jl:685-703) on the focal voltages the device gathered (v - v[src])
                            for (g, i) in enumerate(focal_in_comp)
                                i >= 2 && (voltmatrix[i, cj] = 1 - gat[g, p] / r)
                            end
                        end
                    end
                end
            end
        end

        if use_shortcut
            anchor = findfirst(isequal(csub[1]), points)
            update_shortcut_resistances!(anchor, shortcut, resistances, points, comp)
        end
    end

    use_shortcut && (resistances = shortcut.shortcut_res)
    for i in 1:numpoints
        resistances[i, i] = 0
    end
    r = vcat(vcat(0, orig_pts)', hcat(orig_pts, resistances))
    save_resistances(r, cfg)
    r
end
