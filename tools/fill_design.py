#!/usr/bin/env python3
"""DESIGN.md = docs/DESIGN.md.tmpl with the numbers of the published artifact session (profiles/r06_*) filled in:
    python tools/fill_design.py [r06]
One number per cell, all from ONE session: the bench line, rocprofv3 kernel stats and PMC summaries tools/publish_session.sh copied."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (R, n))
d = json.load(open(P("bench_f16x3.json")))
r, f = d["roofline"], d["f16_path"]
e = d["extra"]["configs"]
pmc = json.load(open(P("bench_f16x3_pmc_eval_split_kernel.json")))["_derived"]
pmc16 = json.load(open(P("bench_f16_pmc_eval_kernel.json")))["_derived"]


def stats(path):
    return {row["Name"].split("(")[0].replace("void ", ""): float(row["AverageNs"]) / 1e3 for row in csv.DictReader(open(path))}


tr = stats(P("train_step_f16x3_kernel_stats.csv"))
inf = stats(P("bench_f16x3_kernel_stats.csv"))
busy, conf = {}, {}
for line in open(P("train_step_f16x3_pmc.txt")):
    m = re.search(r"^(\S.*?)\s+MFMA busy ([0-9.]+) %", line)
    if m:
        busy[m.group(1)] = m.group(2)
    m = re.search(r"^(\S.*?)\s+MFMA.*LDS bank conflicts ([0-9.]+) %", line)
    if m:
        conf[m.group(1)] = m.group(2)
pick = lambda table, key: next(v for k, v in table.items() if key in k)
ms_rocprof = [float(l.split()[0]) for l in open(P("train_step_f16x3_ms.txt"))]
ts, ts16 = e["train_step_fp32_class"], e["train_step"]
eager = e["train_step_torch_eager_gpu_baseline"]["ms_per_step"]
k = lambda v: "%.1f" % (v / 1e3)
o = e["eval_object_loop"]
assert o["same_draws_check"]["u8_images_identical"], "object-loop identity check failed in this session"
c1 = d["cpu_baseline_config1"]
V = {
    "VAL": k(d["value"]), "MS": "%.1f" % d["ms_per_step"], "FRAC": "%.3f" % r["frac"], "ACH": "%.1f" % r["achieved"],
    "EXE": "%.0f" % r["executed_mfma_tflops"], "EXEF": "%.3f" % r["executed_mfma_frac_of_peak"],
    "BUSY": "%.1f" % (100 * r["mfma_busy_frac_measured_in_run"]), "CLK": "%.3f" % r["shader_clock_ghz_during_kernel"],
    "L2": "%.1f" % (100 * pmc["L2_hit_rate"]), "TRAF": "%.1f" % (r["traffic"] / 1e9), "AVG": "%.2f" % r["avg_launch_ms"],
    "ROCAVG": "%.2f" % (pick(inf, "eval_split_kernel") / 1e3),
    "PSNR": "%.1f" % d["psnr_db"], "PSNRF": "%.1f" % d["psnr_db_full_size_vs_f32_hip"],
    "EAG": k(d["torch_eager_gpu_baseline_ref_shape"]["value"]), "SPD": "%.1f" % d["speedup_vs_torch_eager_gpu"],
    "EAGU": k(d["torch_eager_gpu_baseline_unchunked_16384"]["value"]), "SPDU": "%.1f" % d["speedup_vs_torch_eager_gpu_unchunked_16384"],
    "LAT": "%.1f" % d["latency_4096_rays_ms_by_precision"]["f16x3"], "F16LAT": "%.1f" % d["latency_4096_rays_ms_by_precision"]["f16"],
    "F16VAL": k(f["value"]), "F16FRAC": "%.3f" % f["roofline"]["frac"], "F16EXE": "%.0f" % f["roofline"]["executed_mfma_tflops"],
    "F16BUSY": "%.1f" % (100 * pmc16["mfma_busy_frac"]), "F16L2": "%.1f" % (100 * pmc16["L2_hit_rate"]),
    "F16TRAF": "%.1f" % ((pmc16["fabric_read_bytes_per_launch"] + pmc16["hbm_write_bytes_per_launch"]) / 1e9),
    "F16PSNR": "%.1f" % f["psnr_db"], "F16SPD": "%.1f" % (f["value"] / d["torch_eager_gpu_baseline_ref_shape"]["value"]),
    "CPU": "%.0f" % d["cpu_baseline"]["value"], "CPU1": "%.2f" % (c1["value"] / 1e3), "HIP1MS": "%.2f" % c1["hip_same_call_ms"],
    "HIP1": "%.2f" % (c1["hip_same_call_rays_per_s"] / 1e6), "HIP1P": "%.1f" % c1["psnr_db_hip_vs_cpu"],
    "ENC1": "%.2f" % d["encode_ms"]["1_images_ms"], "ENC16": "%.2f" % d["encode_ms"]["16_images_ms"],
    "STEP": "%.2f" % ts["ms_per_step"], "STEPX": "%.1f" % (eager / ts["ms_per_step"]), "EAGER": "%.1f" % eager,
    "STEPR": "%.2f–%.2f" % (min(ms_rocprof), max(ms_rocprof)), "STEPMV": "%.2f" % e["train_step_fp32_class_multiview"]["ms_per_step"],
    "STEPDTU": "%.2f" % e["train_step_fp32_class_dtu"]["ms_per_step"],
    "STEPDTU4": "%.1f" % e["train_step_fp32_class_dtu_batch4"]["ms_per_step"],
    "GRAPH": "%.2f" % ts["hip_graph"]["ms_per_step"], "TWIN": "%.1f" % e["train_step_fp32_class_gemm_per_layer"]["ms_per_step"],
    "F32": "%.1f" % e["train_step_fp32_validation_path"]["ms_per_step"],
    "STEP16": "%.2f" % ts16["ms_per_step"], "STEP16X": "%.1f" % (eager / ts16["ms_per_step"]),
    "EVALT": "%.0f" % pick(tr, "eval_split_kernel<true, false, false, true, false>"), "EVALB": pick(busy, "eval_split_kernel"),
    "BWDT": "%.0f" % pick(tr, "::bwd_split_kernel"), "BWDB": pick(busy, "bwd_split_kernel<"),
    "DWT": "%.0f" % pick(tr, "dw_split_wide_kernel"), "DWB": pick(busy, "dw_split_wide_kernel"),
    "SCT": "%.0f" % pick(tr, "latent_scatter_owner_kernel"), "SEGT": "%.0f" % pick(tr, "scatter_segments_kernel"),
    "SCC": pick(conf, "latent_scatter_owner_kernel"),
    "OBJA": "%.1f" % o["reference_shaped_loop"]["ms_per_object"], "OBJB": "%.1f" % o["render_views_plus_epilogue"]["ms_per_object"],
    "OBJX": "%.3f" % o["speedup"],
}
for name, tag in (("srn_car", "SRN"), ("dtu", "DTU"), ("dtu_9v", "D9")):
    x = e[name]["f16x3"]
    V[tag] = k(x["rays_per_s"]); V[tag + "F"] = "%.3f" % x["frac_of_f16_mfma_peak"]; V[tag + "P"] = "%.0f" % x["psnr_db_vs_cpu_oracle"]
    V[tag + "FOLD"] = "%.2f" % x["fold_ms_both_networks"]; V[tag + "MS"] = "%.2f" % (x["ms_per_call"] / 1e3)
    if "f16" in e[name]:
        V[tag + "16"] = k(e[name]["f16"]["rays_per_s"]); V[tag + "16F"] = "%.3f" % e[name]["f16"]["frac_of_f16_mfma_peak"]
text = open(os.path.join(ROOT, "docs", "DESIGN.md.tmpl")).read()
missing = sorted(set(re.findall(r"@([A-Z0-9]+)@", text)) - set(V))
assert not missing, missing
text = re.sub(r"@([A-Z0-9]+)@", lambda m: V[m.group(1)], text)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md written from profiles/%s_* (%s k rays/s, frac %s, step %s ms)" % (R, V["VAL"], V["FRAC"], V["STEP"]))
