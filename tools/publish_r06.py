"""Copies what tools/collect_r06.sh / collect_gr1t2.sh left under gpurun_out/r06 into profiles/ under the names DESIGN.md cites."""
import glob, json, os, shutil
src, dst = "gpurun_out/r06", "profiles"
pairs = {"bench_rough.json": "r06_bench_n1_rough4096.json", "bench_rough_runs.jsonl": "r06_bench_n1_rough4096_runs.jsonl",
         "bench_driver_window.json": "r06_bench_n1_rough4096_driver_window.json", "bench_flat.json": "r06_bench_n1_flat4096.json",
         "bench_rough_every_step.json": "r06_bench_n1_rough4096_every_step.json", "sweep.jsonl": "r06_bench_n1_rough_sweep.jsonl",
         "bench_full_body_rough4096.json": "r06_bench_n1_full_body_rough4096.json", "bench_full_body_rough16384.json": "r06_bench_n1_full_body_rough16384.json",
         "bench_gr1t2_rough4096.json": "r06_bench_n1_gr1t2_rough4096.json",
         "bench_trimesh.json": "r06_bench_n1_trimesh4096.json", "bench_trimesh8192.json": "r06_bench_n1_trimesh8192.json"}
for a, b in pairs.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for d in glob.glob(src + "/stats_*/"):
    wl = os.path.basename(d.rstrip("/"))[len("stats_"):]
    shutil.copy(os.path.join(d, "b_kernel_stats.csv"), os.path.join(dst, f"r06_kernel_stats_{wl}.csv"))
for f in glob.glob(src + "/pmc_json/r06_pmc_*.json") + glob.glob(src + "/r06_pmc_*.json"):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for b in sorted(pairs.values()):
    for line in open(os.path.join(dst, b)):
        j = json.loads(line); r = j["roofline"]
        print(b, round(j["value"] / 1e6, 2), "M", round(r["kernel_ms"] * 1e3, 2), "us frac", round(r["frac"], 4), "traffic", r["traffic"], "valu", r["valu_issue_frac"] and round(r["valu_issue_frac"], 3))
