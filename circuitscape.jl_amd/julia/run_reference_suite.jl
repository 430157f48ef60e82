# run_reference_suite.jl -- the first thing to run on a machine that HAS Julia, a patched Circuitscape.jl checkout
# (INTEGRATION.md section 2: alias list, enum value, get_solver branch, include("CircuitscapeHIPExt.jl")) and an MI355X
# with libcsgpu.so on the loader path (or ENV["LIBCSGPU"] pointing at it).
#
#   julia --project=<patched Circuitscape.jl> circuitscape.jl_amd/julia/run_reference_suite.jl [--single]
#
# It drives the reference's OWN regression suite through `solver = hip`:
#   1. struct layout of the ccall mirrors against the library (sizes reported by csgpu_default_opts' struct_size field),
#   2. `runtests(solver = "hip", parallel = false)` of test/test_utils.jl:62-140 -- the 54 golden cases (network pairwise /
#      advanced, raster pairwise 1-17, raster advanced 1-6, one-to-all 1-13, all-to-one 1-12: resistances, voltage and
#      current maps, at the reference's tolerances 1e-6 / 1e-4),
#   3. the include / exclude-pairs scenarios of test/issue341.jl with the solver line of every generated INI rewritten,
#   4. (--single) the suite once more with `precision = single` (expect sgVerify17 to miss the helper's 1e-2 criterion with ANY
#      solver: the Float32 regularisation shift of core.jl:161 moves that case's exact solution by 1.0e-2 relative, DESIGN.md
#      section 2 "the fp32 contract"; the Python mirror of this run is tests/helpers.py::check_golden_single_precision).
# The build image of this repository has no Julia (INTEGRATION.md), so this file has never been executed; what it calls is
# mirrored line by line by tests/ (Python host mirror + ctypes) -- tests/test_gpu_golden.py runs the same 54 cases and
# tests/test_issue341.py the same scenarios on the device. tests/test_julia_binding.py checks this script statically
# (the files it includes exist in the reference, the solver alias is one of the patch's).
using Test
using Circuitscape

const CS_ROOT = normpath(joinpath(dirname(pathof(Circuitscape)), ".."))
const CS_TEST = joinpath(CS_ROOT, "test")

@testset "libcsgpu binding" begin
    @test isdefined(Circuitscape, :HIPAMGSolver)                         # the patch is in
    o = Circuitscape.default_opts(16)
    @test o.struct_size == sizeof(Circuitscape.CsgpuOpts)                # C struct and Julia mirror agree on the size
    @test Circuitscape.device_count() >= 1
    st = Circuitscape.CsgpuStats()
    @test sizeof(st) == 128                                              # csgpu_stats (include/csgpu.h): 128 bytes
end

cd(CS_TEST) do
    include(joinpath(CS_TEST, "test_utils.jl"))                          # compute_with, runtests, check_resistances, ...
    clean_output()
    runtests(solver = "hip", parallel = false)                           # blocking ccalls: no task fan-out over them
    if "--single" in ARGS
        runtests(solver = "hip", precision = "single", parallel = false)
    end

    # test/issue341.jl writes its INIs with `solver = cg+amg`; the same scenarios with the solver swapped. compute(path)
    # is shadowed for the duration of the include so that every job of the file runs through the HIP solver.
    @testset "Issue 341: included pairs (solver = hip)" begin
        mod = Module(:Issue341HIP)
        Core.eval(mod, :(using Test, DelimitedFiles))
        Core.eval(mod, :(import Circuitscape))
        Core.eval(mod, :(compute(path::String) = Circuitscape.compute(
            let d = Dict{String,String}(Circuitscape.parse_config(path)); d["solver"] = "hip"; d end)))
        Core.eval(mod, :(using Circuitscape: parse_config))
        src = replace(read(joinpath(CS_TEST, "issue341.jl"), String), "using Circuitscape, Test, DelimitedFiles" => "")
        Base.include_string(mod, src, "issue341.jl")
    end
    clean_output()
end
