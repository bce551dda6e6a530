#!/bin/bash
# Same-session A/B of the bench under two OVM_TUNE settings: ab_bench.sh "<tune A>" "<tune B>" [rounds]; prints ms_per_step per run.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
A="$1"; B="$2"; N=${3:-2}
for i in $(seq 1 $N); do
  for T in "$A" "$B"; do
    OVM_TUNE="$T" timeout -k 10 150 python3 $R/bench.py --no-alt --no-cpu-baseline --steps 40 --warmup 8 > $O/ab_tmp.json 2> $O/ab_tmp.err || { echo "FAILED tune=[$T]"; tail -3 $O/ab_tmp.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/ab_tmp.json'));print('tune=[$T]', d['ms_per_step'], 'ms', d['value'], 'img/s', 'det alone', d['kernels']['gdino_network_alone']['ms_per_forward'])"
  done
done
