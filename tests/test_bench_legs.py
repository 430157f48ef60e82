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
