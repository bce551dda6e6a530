#!/bin/bash
# Same-session A/B of the bench under environment settings: ab_env.sh ROUNDS "ENV1=a ENV2=b" "ENV1=c" ...; prints ms_per_step per run.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
N=$1; shift
for i in $(seq 1 $N); do
  for E in "$@"; do
    env $E timeout -k 10 150 python3 $R/bench.py --no-alt --no-cpu-baseline --steps 40 --warmup 8 > $O/ab_tmp.json 2> $O/ab_tmp.err || { echo "FAILED env=[$E]"; tail -3 $O/ab_tmp.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/ab_tmp.json'));print('env=[$E]', d['ms_per_step'], 'ms', d['value'], 'img/s', 'attn', d['kernels']['attn'])"
  done
done
