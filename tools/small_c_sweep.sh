#!/bin/bash
# window size of small MSMs without window tables (table not resident: 24 < n < 1024 host-pointer calls), BBGPU_PLAIN_C sweep, and the fused scan A/B at 2^16
for c in 0 5 6 7 8 9 10; do echo "== BBGPU_PLAIN_C=$c (0 = default lg n - 4)"; if [ $c = 0 ]; then python tools/small_breakdown.py; else BBGPU_PLAIN_C=$c python tools/small_breakdown.py; fi 2>&1 | grep "level 0"; done
for f in 1 0 1 0; do echo "== BBGPU_SORT_SCAN_FUSED=$f"; BBGPU_SORT_SCAN_FUSED=$f python tools/msm_ab.py 2>&1 | grep "2^16, 1[0-9] of" ; done
