#!/bin/bash
# Regenerates every file under profiles/ that the roofline numbers come from, from the CURRENT binary, in one gpurun call:
#   tools/collect_profiles.sh <tag>        (tag = r02 ...; outputs land in gpurun_out/prof_<tag>/, summaries are copied by hand)
# Counter passes carry --pmc only (no tracing domains), one counter group per pass, as MI355X_MICROARCH.md prescribes.
set -o pipefail
#   tools/collect_profiles.sh <tag> 1 | 2   the traces and counter passes (1) or the tool runs behind them (2) alone: one gpurun call holds 20 minutes
TAG=${1:-r05}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != 2 ]; then
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plonk --no-boundary"
BENCH_PMC="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plonk --no-boundary"
echo "== kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
echo "== FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH_PMC > $OUT/fetch.log 2>&1 || exit 1
echo "== WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH_PMC > $OUT/write.log 2>&1 || exit 1
echo "== calibration FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $ROOT/tools/ubench/ubench_traffic > $OUT/cal_fetch.log 2>&1 || exit 1
echo "== calibration WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- $ROOT/tools/ubench/ubench_traffic > $OUT/cal_write.log 2>&1 || exit 1
echo "== 2^22 transforms FETCH_SIZE / WRITE_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ntt22_fetch -- python3 $ROOT/tools/ntt_only.py 22 > $OUT/ntt22_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ntt22_write -- python3 $ROOT/tools/ntt_only.py 22 > $OUT/ntt22_write.log 2>&1 || exit 1
echo "== SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-plonk --no-boundary > $OUT/sq.log 2>&1 || exit 1
echo "== prover trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plonk -- python3 $ROOT/tools/plonk_bench.py --reps 10 --no-reference > $OUT/plonk.log 2>&1 || exit 1
echo "== latency traces"
for L in 20 16; do rocprofv3 --kernel-trace --output-format csv -d $OUT/lat$L -- python3 $ROOT/tools/latency_run.py $L > $OUT/lat$L.log 2>&1 || exit 1; done
cd $ROOT
echo "== summaries"
python tools/timeline.py $OUT/trace 2 > $OUT/timeline_steady_state.txt
python tools/window_timeline.py $OUT/plonk 0.7 2700 > $OUT/prover_timeline.txt
for L in 20 16; do echo "== one 2^$L-point MSM at a time (tools/latency_run.py): start, duration, gap (us), kernel"; python tools/latency_timeline.py $OUT/lat$L; done > $OUT/latency_timeline.txt
rm -rf $OUT/lat20 $OUT/lat16
python tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/cal_fetch $OUT/cal_write $OUT/${TAG}_pmc_traffic.json $OUT/ntt22_fetch $OUT/ntt22_write
python tools/pmc_sq_summary.py $OUT/sq $OUT/${TAG}_pmc_sq.json
python tools/acc_spacing.py $OUT/trace | tee $OUT/acc_spacing.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bench_steps10.csv
cp $(find $OUT/plonk -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_plonk_prover_2e16.csv
$ROOT/tools/ubench/ubench_traffic > $OUT/ubench_traffic.txt 2>&1
# keep the merged scratch small: the raw traces of the counter passes are not needed once summarised
rm -rf $OUT/fetch $OUT/write $OUT/cal_fetch $OUT/cal_write $OUT/sq $OUT/ntt22_fetch $OUT/ntt22_write
find $OUT/trace $OUT/plonk -name "*kernel_trace.csv" -size +20M -delete
fi
cd $ROOT
[ "$PART" = 1 ] && exit 0
echo "== shard prediction"
python tools/shard_sim.py $OUT/${TAG}_shard_prediction.json > $OUT/shard_sim.txt 2>&1; tail -6 $OUT/shard_sim.txt
python tools/ntt_sizes.py > $OUT/ntt_sizes.txt 2>&1
python tools/msm_ab.py > $OUT/msm_ab.txt 2>&1
python tools/small_sizes.py > $OUT/small_sizes.txt 2>&1
python tools/exchange_cost.py > $OUT/exchange_cost.txt 2>&1
python tools/host_finish_cost.py > $OUT/host_finish_cost.txt 2>&1
python tools/share_ab.py 8 4 2 > $OUT/share_ab.txt 2>&1
python tools/point_share_ab.py 8 4 2 > $OUT/point_share_ab.txt 2>&1
python tools/issue_cost.py > $OUT/issue_cost.txt 2>&1
python tools/host_step_cost.py > $OUT/host_step_cost.txt 2>&1
python tools/msm_big.py 20 21 22 24 > $OUT/msm_big.txt 2>&1
for g in 1048576 2097152; do python tools/plonk_bench.py --gates $g --reps 5 --no-reference; done > $OUT/plonk_big.txt 2>&1
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq8 -- python3 $ROOT/tools/share_run.py points 8 2 > $OUT/sq8.log 2>&1 ) && python tools/pmc_sq_summary.py $OUT/sq8 $OUT/${TAG}_pmc_sq_share8_points.json > /dev/null; rm -rf $OUT/sq8
python tools/skewed_stages.py > $OUT/skewed_stages.txt 2>&1
python tools/boundary_ab.py > $OUT/boundary_ab.txt 2>&1
BBGPU_HOST_MSM_SPLIT=1 python tools/boundary_ab.py >> $OUT/boundary_ab.txt 2>&1
python tools/validate_ab.py >> $OUT/boundary_ab.txt 2>&1
python tools/ntt_ramp.py 500 > $OUT/ntt_ramp.txt 2>&1
python tools/acc_ab.py > $OUT/acc_ab.txt 2>&1
$ROOT/tools/ubench/ubench_pcie > $OUT/pcie.txt 2>&1
export OMP_NUM_THREADS=16
for w in 0 2; do BB_WARM_PROOFS=$w BBGPU_SHIM_PROFILE=$OUT/shim_profile_w$w.json oracle/_ref/plonk_gpu prove 65536 > /dev/null 2>> $OUT/shim_profile.log; done
BB_WARM_PROOFS=1 oracle/_ref/plonk_cpu prove 65536 > /dev/null 2>> $OUT/shim_profile.log
echo "== plain bench line"
python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
tail -c 400 $OUT/${TAG}_bench_line.json
