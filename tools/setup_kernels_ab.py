#!/usr/bin/env python
"""A/B of the lattice set-up kernels (round 6): the level-1 Galerkin product from LDS-staged fine columns against the plain
form (CSGPU_GALERKIN_PLAIN=1) and lattice_ap_q_kernel at 256 / 128 / 64 cells per workgroup (CSGPU_APQ_NT). Each variant runs in
a child process (the debug overrides are read when a handle's options are resolved); prints set-up device time, the
iteration counts and a digest of the resistances -- which must be the same bits in every variant.
usage: setup_kernels_ab.py SIZE [holes] [--emu] [--precond fp32]"""
import hashlib, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(size, holes, emu, precond):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    lib.load(os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so") if emu else os.environ.get("CSGPU_LIB"))
    rng = np.random.default_rng(5)
    g = np.exp(0.3 * rng.standard_normal((size, size + 7 if emu else size)))
    if holes > 0:
        g[rng.random(g.shape) < holes] = 0.0
    best = None
    for rep in range(1 if emu else 3):
        t0 = time.time()
        h = lib.raster_setup(g, lib.default_opts(batch=16, precond_bytes=4 if precond == "fp32" else 0))
        wall = time.time() - t0
        info = h.info
        dev = info["setup_ms"] + info["upload_ms"]
        if best is None or (dev is not None and dev < best[0]):
            best = (dev, wall)
        if rep < 2 and not emu:
            h.close()
    n = info["n"]
    pts = rng.choice(n, size=8, replace=False)
    R, _, _, st = h.solve_pairs([int(p) for p in pts[:4]], [int(p) for p in pts[4:]])
    out = {"size": size, "holes": holes, "precond": precond, "galerkin": "staged" if os.environ.get("CSGPU_GALERKIN_STAGED") else "plain",
           "apq_nt": int(os.environ.get("CSGPU_APQ_NT", "256")), "raster_transpose": 0 if os.environ.get("CSGPU_NO_RASTER_TRANSPOSE") else 1,
           "setup_ms": info["setup_ms"], "upload_ms": info["upload_ms"], "setup_device_ms": best[0], "setup_wall_s": best[1],
           "levels": info["levels"], "iters": st["total_iters"], "not_converged": st["not_converged"],
           "digest": hashlib.sha1(np.ascontiguousarray(R).tobytes()).hexdigest()[:16]}
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), float(sys.argv[3]), sys.argv[4] == "1", sys.argv[5])
        sys.exit(0)
    size = int(sys.argv[1])
    holes = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2][0].isdigit() else 0.0
    emu = "--emu" in sys.argv
    precond = "fp32" if "fp32" in sys.argv else "same"
    variants = [{}, {"CSGPU_GALERKIN_STAGED": "1"}, {"CSGPU_GALERKIN_STAGED": "1", "CSGPU_APQ_NT": "128"},
                {"CSGPU_GALERKIN_STAGED": "1", "CSGPU_APQ_NT": "64"}]
    if "--transpose" in sys.argv:   # (second A/B of the round: the raster kernels on a column-major copy of the raster)
        variants = [{}, {"CSGPU_NO_RASTER_TRANSPOSE": "1"}, {}, {"CSGPU_NO_RASTER_TRANSPOSE": "1"}]
    for v in variants:
        env = dict(os.environ)
        env.update(v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(size), str(holes), "1" if emu else "0", precond],
                           env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            sys.stdout.write(json.dumps({"variant": v, "rc": r.returncode, "err": r.stderr[-400:]}) + "\n")
        sys.stdout.flush()
