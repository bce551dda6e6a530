import sys, os, json, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
# bench.py builds its cfg through tests/common.build_cfg; co-run is a config key, so drive it through an env hook the bench honours
for corun in ("0", "1", "0", "1"):
    env = dict(os.environ, OVM_BENCH_CORUN=corun)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "4", "--no-cpu-baseline", "--no-alt"], capture_output=True, text=True, env=env, timeout=600)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print("corun", corun, d["value"], d["ms_per_step"], d["kernels"]["attn"]["ms_per_step"], d["kernels"]["gdino_network"]["ms_per_step"], flush=True)
