"""bench.py's single-GPU legs on the CPU emulator build (kernel logic only; test infrastructure): the line must carry the
NODATA leg added in round 4 with the fields a reader needs. (--precision single: the configs[3] / configs[4] legs, which
are sized for a GPU, only run on the fp64 line.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_carries_the_nodata_leg(emu_lib):
    env = dict(os.environ, CSGPU_LIB=os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "120", "--steps", "1", "--warmup", "1",
                          "--batch", "8", "--cpu-sample", "0", "--precision", "single", "--host-csr", "0"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    leg = line["nodata15"]
    assert "failed" not in leg, leg
    for key in ("value", "ms_per_16_pairs", "iters_mean", "iters_max", "max_relres", "not_converged", "nodes",
                "giant_component_nodes", "lattice_period", "levels", "setup_s", "nodata_fraction"):
        assert key in leg, key
    assert leg["not_converged"] == 0 and leg["lattice_period"] == 120 and leg["nodata_fraction"] == 0.15
    assert 0.8 * 120 * 120 < leg["nodes"] < 0.9 * 120 * 120 and leg["value"] > 0
    assert line["value"] > 0 and "shortcut" in line and "with_voltages" in line
    # BASELINE configs[2] as a job: the 100-pair list in one call, 12 full batches of 8 + 4
    job = line["job_100_pairs"]
    assert "failed" not in job, job
    assert job["pairs"] == 100 and job["batches"] == "12 x 8 + 4" and job["not_converged"] == 0
    assert abs(line["value_job"] - 100 / job["job_s"]) < 1e-9 and line["job_100_pairs_s"] == job["job_s"]


def test_every_leg_of_the_fp64_line_carries_an_oracle_figure(emu_lib):
    """nodata15, config3_fp32, config4_network (+ the geometric network that coarsens) each print a
    `parity` object with max_rel_err / tolerance / ok, computed against the tight oracle (scipy Jacobi-CG for the Jacobi-
    preconditioned expander) -- here at toy sizes on the emulator build; the driver runs the real sizes."""
    env = dict(os.environ, CSGPU_LIB=os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "120", "--steps", "1", "--warmup", "1",
                          "--batch", "8", "--cpu-sample", "90", "--leg-sample", "96", "--network-n", "4000", "--geometric-n",
                          "4000", "--host-csr", "0"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["parity"]["ok"] and line["parity"]["max_rel_err_vs_oracle"] < 1e-6
    for leg, tol in (("nodata15", 1e-6), ("config3_fp32", 1e-4), ("config4_network", 1e-6), ("network_geometric", 1e-6)):
        assert "failed" not in line[leg], (leg, line[leg])
        par = line[leg]["parity"]
        assert "failed" not in par, (leg, par)
        assert par["tolerance"] == tol and par["ok"] and par["max_rel_err"] < tol, (leg, par)
    assert line["config4_network"]["levels"] == 1 and line["network_geometric"]["levels"] >= 3
    # device time and a CSR-SpMM roofline in both network legs
    for leg in ("config4_network", "network_geometric"):
        roof = line[leg]["roofline"]
        assert roof["bound"] == "hbm" and roof["algorithmic_bytes_per_launch"] > 0 and roof["launches_timed"] > 0, (leg, roof)
        assert line[leg]["value_device"] > 0
    assert line["config4_network"]["solve_device_s_all_sources"] > 0 and line["config4_network"]["cum_current_sum"] > 0
    assert line["nodata15"]["parity"]["lattice_period"] == 96
    assert set(line["leg_seconds"]) >= {"nodata15", "config3_fp32", "config4_network", "cpu_child", "leg_parity_gpu"}
