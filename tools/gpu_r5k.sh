#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5k
mkdir -p $O
timeout 400 python tools/soak.py 1500 gomoku 13 4096 fp32 > $O/soak_gomoku13.json 2> $O/soak_gomoku13.err; echo "soak c2 rc=$?" > $O/status.txt
timeout 400 python tools/soak.py 2000 go 9 4096 fp32 > $O/soak_go9.json 2> $O/soak_go9.err; echo "soak c3 rc=$?" >> $O/status.txt
cat $O/status.txt; tail -1 $O/soak_gomoku13.json; tail -1 $O/soak_go9.json; tail -3 $O/soak_gomoku13.err
