#!/bin/bash
# Round 3, last GPU call: the Dirichlet-masked solves after the stopping-rule fix and the coarsest-level correction --
# their GPU tests, one-to-all on the shared hierarchy (2000^2, 16 points) and network config 5 with / without the
# correction (CSGPU_NO_DIRICHLET_COARSE=1).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
rm -rf $OUT; mkdir -p $OUT
timeout 420 python -m pytest tests -m gpu -x -q -k "grounded or region or onetoall or alltoone or dirichlet or advanced or Network or network or shared" > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
timeout 200 python tools/onetoall_bench.py 2000 > $OUT/onetoall.jsonl 2> $OUT/onetoall.err; tail -1 $OUT/onetoall.jsonl | cut -c1-700
CSGPU_NO_DIRICHLET_COARSE=1 timeout 200 python tools/onetoall_bench.py 2000 > $OUT/onetoall_off.jsonl 2> $OUT/onetoall_off.err; tail -1 $OUT/onetoall_off.jsonl | cut -c1-700
NFOCAL=16 timeout 240 python tools/network_bench.py 5000000 16 --shared > $OUT/net5.jsonl 2> $OUT/net5.err; tail -1 $OUT/net5.jsonl | cut -c1-700
