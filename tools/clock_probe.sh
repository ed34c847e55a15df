#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/acc_ab.py > /tmp/acc.txt 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8; echo --; sleep 1.5; done
wait $PID
grep accumulate /tmp/acc.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
