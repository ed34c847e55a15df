#!/bin/bash
# one-box A/B of the fused scan launch (BBGPU_SORT_SCAN_FUSED) at 2^16 points, and the small host-pointer MSMs with tables that are not resident
for f in 1 0 1 0; do echo "== BBGPU_SORT_SCAN_FUSED=$f"; BBGPU_SORT_SCAN_FUSED=$f python tools/msm_ab.py 2>&1 | grep "2^16, 1[0-9] of"; done
python tools/small_breakdown.py 2>&1 | grep "level 0"
python tools/small_sizes.py
