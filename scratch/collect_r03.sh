#!/bin/bash
# Round-3 measurements on one MI355X box: bench lines, rocprofv3 kernel-trace statistics of the bench command, and the PMC passes
# (SQ counters; FETCH_SIZE; WRITE_SIZE - separate passes, counters only) over the SAME workload the bench line prints
# (BASELINE configs[1], ROIHeads3DGDINO). Outputs under gpurun_out/r03/; scratch/summarise_r03.py turns them into profiles/r03/.
set -o pipefail
# every GPU step stops the script when it fails or is killed at its limit: nothing further is started on the box after that
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
PART=${1:-all}      # "bench": the bench lines; "prof": kernel trace + PMC passes (two gpurun calls: together they exceed one call's limit)
if [ "$PART" != "prof" ]; then
timeout -k 10 500 $B > $O/bench_final.json 2> $O/bench_final.err || { echo "bench FAILED rc=$?"; exit 1; }; echo "bench ok"
timeout -k 10 200 $B --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_oracle2d.json 2> $O/bench_oracle2d.err || { echo "oracle2d FAILED rc=$?"; exit 1; }; echo "oracle2d ok"
# the route the fork's --eval-only takes without oracle boxes (RPN -> box head), ViT-B as in configs/OVMono3D_dinov2_SFP.yaml and nohup.out:939, with parity
timeout -k 10 400 $B --proposals rpn --model vitb14 --no-alt > $O/bench_rpn_vitb14.json 2> $O/bench_rpn_vitb14.err || { echo "rpn vitb FAILED rc=$?"; exit 1; }; echo "rpn vitb ok"
timeout -k 10 400 $B --proposals rpn --no-alt > $O/bench_rpn_vitl14.json 2> $O/bench_rpn_vitl14.err || { echo "rpn vitl FAILED rc=$?"; exit 1; }; echo "rpn vitl ok"
# BASELINE configs 3 / 4 / 5 at their per-GPU batch (8), parity over every image of the batch
timeout -k 10 600 $B --proposals oracle2d --no-alt --batch 8 --steps 10 --warmup 2 > $O/bench_c3_b8.json 2> $O/bench_c3_b8.err || { echo "c3 b8 FAILED rc=$?"; exit 1; }; echo "c3 b8 ok"
timeout -k 10 600 $B --tower clip --model ViT-B-16 --canvas 1024 --net-res 608 --proposals oracle2d --batch 8 --steps 10 --warmup 2 --no-alt > $O/bench_c4_clip_b8.json 2> $O/bench_c4_clip_b8.err || { echo "c4 b8 FAILED rc=$?"; exit 1; }; echo "c4 b8 ok"
timeout -k 10 900 $B --canvas 1036 --net-res 1024 --batch 8 --proposals oracle2d --no-alt --steps 5 --warmup 1 > $O/bench_c5_b8.json 2> $O/bench_c5_b8.err || { echo "c5 b8 FAILED rc=$?"; exit 1; }; echo "c5 b8 ok"
# the other towers of the reference's configs at batch 1, given boxes (parity-test cases; throughput for the record)
timeout -k 10 200 $B --tower clip --model ViT-B-16 --canvas 1024 --net-res 608 --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_clip_b1.json 2> $O/bench_clip_b1.err || { echo "clip FAILED rc=$?"; exit 1; }; echo "clip ok"
fi
if [ "$PART" != "bench" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_under_rocprof.json 2> $O/kt.err || { echo "kernel-trace FAILED rc=$?"; exit 1; }; echo "kernel-trace ok"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_sq -o sq -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_sq.json 2> $O/pmc_sq.err || { echo "pmc sq FAILED rc=$?"; exit 1; }; echo "pmc sq ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_fetch.json 2> $O/pmc_fetch.err || { echo "pmc fetch FAILED rc=$?"; exit 1; }; echo "pmc fetch ok"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_write.json 2> $O/pmc_write.err || { echo "pmc write FAILED rc=$?"; exit 1; }; echo "pmc write ok"
# the CSVs are large: keep per-kernel means only
for d in pmc_sq pmc_fetch pmc_write; do
  f=$(find $O/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/scratch/pmc_summary.py "$f" "ovm|_GLOBAL__N" > $O/$d.summary.json && rm -rf $O/$d
done
f=$(find $O/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/kt
fi
ls -la $O
