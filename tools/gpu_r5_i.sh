#!/bin/bash
# Round 5: strong scaling from COLD per-rank pieces: every rank of an 8-GPU job is a fresh process, so
# the honest single-GPU pieces are the cold first runs -- the whole job on one GPU and one rank's share, each in its own
# process; fp64 100 pairs (configs[2]) and fp32 1000 pairs (configs[3]).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5i
rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py --scaling strong --pairs 100 > $OUT/strong_100_fp64.json 2> $OUT/err.log
timeout 300 python bench.py --scaling strong --pairs 13 > $OUT/strong_13_fp64.json 2>> $OUT/err.log
timeout 300 python bench.py --scaling strong --pairs 1000 --precision single > $OUT/strong_1000_fp32.json 2>> $OUT/err.log
timeout 300 python bench.py --scaling strong --pairs 125 --precision single > $OUT/strong_125_fp32.json 2>> $OUT/err.log
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5i/"
def rd(f): return json.loads(open(o + f).read().strip().splitlines()[-1])
res = {}
for tag, full, share, n in (("fp64_100_pairs", "strong_100_fp64.json", "strong_13_fp64.json", 8), ("fp32_1000_pairs", "strong_1000_fp32.json", "strong_125_fp32.json", 8)):
    a, b = rd(full), rd(share)
    res[tag] = {"T1_cold_s": a["job_cold_s"], "T1_warm_s": a["job_s"], "share_cold_s": b["job_cold_s"], "share_warm_s": b["job_s"],
                "share_pairs": b["config"]["workload"], "speedup_8gpu_from_cold_pieces": a["job_cold_s"] / b["job_cold_s"],
                "speedup_8gpu_from_warm_pieces": a["job_s"] / b["job_s"], "setup_s_rank0": a["rank_setup_s"]}
json.dump(res, open(o + "strong_cold_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
tail -n 3 $OUT/err.log
