#!/usr/bin/env python3
"""Turn a PMC pass (tools/gpu_pmc.sh -> gpurun_out/pmc_bench/pmc_by_kernel.json) into the entry of
profiles/pmc_traffic.json that bench.py reports as `roofline.traffic`: HBM bytes per launch of the fine-level CG product =
FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section: the counter's unit is 64 B on this part while rocprofv3
scales it as 32 B; verified in every pass on a streaming kernel of known size) + WRITE_SIZE, both KiB. The entry carries the
hash of the kernel sources it was measured with (bench.kernel_source_hash()); bench.py prints null when the hash no longer
matches.

usage: python tools/pmc_update.py <pmc_by_kernel.json> <key> "<kernel name prefix>" <algorithmic bytes per launch>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    src, key, kprefix, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    by = json.load(open(src))
    hits = [k for k in by if k.startswith(kprefix)]
    assert len(hits) == 1, (kprefix, hits)
    c = by[hits[0]]
    fetch = c["FETCH_SIZE"].get("mean_fullsize", c["FETCH_SIZE"]["max"])
    write = c["WRITE_SIZE"].get("mean_fullsize", c["WRITE_SIZE"]["max"])
    traffic = (2.0 * fetch + write) * 1024.0
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    d[key] = {"kernel": hits[0], "FETCH_SIZE_KiB_per_real_launch": fetch, "WRITE_SIZE_KiB_per_real_launch": write,
              "launches": c["FETCH_SIZE"].get("n_fullsize", c["FETCH_SIZE"]["n"]),
              "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
              "traffic_over_algorithmic": traffic / alg, "kernel_src_sha16": bench.kernel_source_hash(),
              "kernel_sources": list(bench.KERNEL_SOURCES),
              "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE exact; separate "
                            "--pmc passes with --kernel-trace only (tools/gpu_pmc.sh)"}
    json.dump(d, open(path, "w"), indent=1)
    print(key, json.dumps(d[key]))


if __name__ == "__main__":
    main()
