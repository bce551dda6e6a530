#!/bin/bash
# Round-2 measurements on one MI355X box: bench lines, rocprofv3 kernel-trace statistics of the bench command, and the PMC passes
# (SQ counters; FETCH_SIZE; WRITE_SIZE - separate passes, counters only) over the SAME workload the bench line prints
# (BASELINE configs[1], ROIHeads3DGDINO). Outputs under gpurun_out/r02/; scratch/summarise_r02.py turns them into profiles/r02/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout -k 10 500 $B > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout -k 10 200 $B --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_oracle2d.json 2> $O/bench_oracle2d.err; echo "oracle2d rc=$?"
timeout -k 10 300 $B --proposals oracle2d --no-alt --no-cpu-baseline --batch 16 --steps 10 > $O/bench_oracle2d_b16.json 2> $O/bench_oracle2d_b16.err; echo "b16 rc=$?"
# the other towers of the reference's configs at batch 1, given boxes (parity-test cases; throughput for the record)
timeout -k 10 200 $B --tower clip --model ViT-B-16 --canvas 1024 --net-res 608 --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_clip_b1.json 2> $O/bench_clip_b1.err; echo "clip rc=$?"
timeout -k 10 200 $B --tower mae --model facebook/vit-mae-base --canvas 1024 --net-res 608 --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_mae_b1.json 2> $O/bench_mae_b1.err; echo "mae rc=$?"
timeout -k 10 200 $B --tower midas --model DPT_Large --canvas 1024 --net-res 608 --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_midas_b1.json 2> $O/bench_midas_b1.err; echo "midas rc=$?"
timeout -k 10 200 $B --tower sam --model vit_b --canvas 1024 --net-res 608 --proposals oracle2d --no-alt --no-cpu-baseline > $O/bench_sam_b1.json 2> $O/bench_sam_b1.err; echo "sam rc=$?"
timeout -k 10 300 $B --tower clip --model ViT-B-16 --canvas 1024 --net-res 608 --proposals oracle2d --batch 32 --steps 5 --warmup 2 --no-alt > $O/bench_clip_b32.json 2> $O/bench_clip_b32.err; echo "clip b32 rc=$?"
timeout -k 10 300 $B --canvas 1036 --net-res 1024 --batch 64 --proposals oracle2d --no-alt --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_c5_b64.json 2> $O/bench_c5_b64.err; echo "c5 rc=$?"
timeout -k 10 200 python3 $R/scratch/bench_jpeg.py > $O/bench_jpeg.json 2> $O/bench_jpeg.err; echo "jpeg rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kernel-trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_sq -o sq -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_sq.json 2> $O/pmc_sq.err; echo "pmc sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_write.json 2> $O/pmc_write.err; echo "pmc write rc=$?"
# the CSVs are large: keep per-kernel means only
for d in pmc_sq pmc_fetch pmc_write; do
  f=$(find $O/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/scratch/pmc_summary.py "$f" "ovm|_GLOBAL__N" > $O/$d.summary.json && rm -rf $O/$d
done
f=$(find $O/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/kt
ls -la $O
