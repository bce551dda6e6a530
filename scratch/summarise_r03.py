"""gpurun_out/r03/ (scratch/collect_r03.sh) -> profiles/r03/: bench lines, kernel statistics, PMC summaries with derived figures."""
import json, os, shutil
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
O, P = os.path.join(R, "gpurun_out", "r03"), os.path.join(R, "profiles", "r03")
os.makedirs(P, exist_ok=True)
def last(f): return json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
for src, dst in (("bench_final.json", "bench_r03_final.json"), ("bench_oracle2d.json", "bench_r03_oracle2d.json"),
                 ("bench_rpn_vitb14.json", "bench_r03_rpn_vitb14.json"), ("bench_rpn_vitl14.json", "bench_r03_rpn_vitl14.json"),
                 ("bench_c3_b8.json", "bench_r03_c3_vitl896_b8.json"), ("bench_c4_clip_b8.json", "bench_r03_c4_clip1024_b8.json"),
                 ("bench_c5_b8.json", "bench_r03_c5_vitl1036_b8.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("bench_clip_b1.json", "bench_r03_clip_b1.json")):
    if not os.path.exists(os.path.join(O, src)):
        continue
    json.dump(last(src), open(os.path.join(P, dst), "w"), indent=1)
if os.path.exists(os.path.join(O, "bench_jpeg.json")):
    shutil.copy(os.path.join(O, "bench_jpeg.json"), os.path.join(P, "bench_r03_jpeg.jsonl"))
# kernel statistics: top 60 rows
rows = open(os.path.join(O, "kernel_stats.csv")).read().splitlines()
open(os.path.join(P, "kernel_stats_f16x3_vitl14_gdino_T4097_b1.csv"), "w").write("\n".join(rows[:61]) + "\n")
CMD = "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt   (default workload: ROIHeads3DGDINO, the configuration the bench line prints)"
sq = json.load(open(os.path.join(O, "pmc_sq.summary.json")))
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS "
               "GRBM_GUI_ACTIVE --output-format csv -- " + CMD + "; counters only (no trace domains); per-dispatch means. mfma_util = MFMA busy cycles / "
               "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (fractions of the wave cycles). Dispatches are "
               "serialised under counter collection, so the figures are uncontended even though the detector shares the chip in the timed bench.",
       "kernels": {}}
for k, v in sq.items():
    if v.get("SQ_WAVE_CYCLES", 0) < 5e6:
        continue
    d = {c: round(x, 1) for c, x in v.items()}
    d["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0), 3)
    d["frac_wait_any"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    d["frac_wait_inst"] = round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    d["frac_valu"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3)
    out["kernels"][k] = d
json.dump(out, open(os.path.join(P, "pmc_sq_f16x3_vitl14_T4097_b1.json"), "w"), indent=1)
f = json.load(open(os.path.join(O, "pmc_fetch.summary.json"))); w = json.load(open(os.path.join(O, "pmc_write.summary.json")))
T, D = 4097, 1024
alg = {"attn_kernel": 4 * T * D * 4, "gemm256_kernel<2": (T * D + 4 * D * D) * 4 + T * 4 * D * 4, "gemm256_kernel<3": (T * D + 3 * D * D) * 4 + T * 3 * D * 4,
       "gemm_ws_kernel<3, 32, 3, 1, 0, true>": ((T * D + D * D) * 4 + T * D * 4 * 2 + (T * 4 * D + 4 * D * D) * 4 + T * D * 4 * 2) // 2}
tr = {"note": "separate passes: rocprofv3 --pmc FETCH_SIZE --output-format csv -- " + CMD + " ; same with WRITE_SIZE (the two together exceed the TCC counter slots). "
              "Counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
              "(MI355X_MICROARCH.md, HBM section). Per-dispatch means. Memory-side requests include Infinity-Cache hits. algorithmic_bytes: operands read "
              "once + outputs written once in split fp16 (4 B per element).",
      "workload": {"model": "vitl14", "canvas": 896, "precision": "f16x3", "batch": 1, "proposals": "gdino"}, "kernels": {}}
for k, v in f.items():
    if v.get("FETCH_SIZE", 0) < 5000 or k not in w:
        continue
    d = {"dispatches": v["dispatches"], "FETCH_SIZE_KiB": round(v["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(w[k]["WRITE_SIZE"], 1),
         "traffic_bytes_per_launch": int((2 * v["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024)}
    for a, b in alg.items():
        if a in k:
            d["algorithmic_bytes_per_launch"] = b
    tr["kernels"][k] = d
json.dump(tr, open(os.path.join(P, "pmc_traffic_f16x3_vitl14_T4097_b1.json"), "w"), indent=1)
print(sorted(os.listdir(P)))
for k, d in out["kernels"].items():
    if d["mfma_util"] > 0.05: print(f"{d['mfma_util']:.3f} wait_any {d['frac_wait_any']:.2f} wait_inst {d['frac_wait_inst']:.2f}  {k[:80]}")
for k, d in tr["kernels"].items():
    if "algorithmic_bytes_per_launch" in d: print(k[:60], d["traffic_bytes_per_launch"] / 1e6, "MB vs algorithmic", d["algorithmic_bytes_per_launch"] / 1e6)
