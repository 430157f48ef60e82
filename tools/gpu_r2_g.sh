#!/bin/bash
# Round 2, call G: host polling / graph replay at the large size, multi-device handle on one GPU, strong-scaling mode, default bench incl. CPU leg.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/default.json 2> $OUT/default.err
timeout 200 $B --opt check_every=2 > $OUT/check2.json 2> $OUT/check2.err
timeout 200 $B --opt check_every=4 > $OUT/check4.json 2> $OUT/check4.err
timeout 200 $B --opt check_every=2 --opt use_graph=1 > $OUT/check2_graph.json 2> $OUT/check2_graph.err
timeout 200 $B --opt check_every=4 --opt use_graph=1 > $OUT/check4_graph.json 2> $OUT/check4_graph.err
timeout 200 $B --opt nu_coarse=2 > $OUT/nuc2.json 2> $OUT/nuc2.err
timeout 200 $B --opt nu_coarse=1 > $OUT/nuc1.json 2> $OUT/nuc1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2g/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f frac %.3f iters %.2f relres %.2e setup %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"], d["iters_mean"], d["max_relres"], d["setup_s"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
timeout 300 python bench.py --scaling strong --pairs 100 > $OUT/strong100.json 2> $OUT/strong100.err; tail -c 1500 $OUT/strong100.json; tail -3 $OUT/strong100.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -c 6000 $OUT/bench_full.json; tail -3 $OUT/bench_full.err
