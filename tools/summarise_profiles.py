"""Copy the artefacts tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/ (tracked).
usage: python tools/summarise_profiles.py [tag]"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/stats/b_kernel_stats.csv", f"{dst}/{tag}_kernel_stats_rough4096.csv")
for a, b in (("bench_rough.json", "bench_n1_rough4096.json"), ("bench_flat.json", "bench_n1_flat4096.json"), ("sweep.jsonl", "bench_n1_rough_sweep.jsonl"), ("bench_rough_runs.jsonl", "bench_n1_rough4096_runs.jsonl")):
    shutil.copy(f"{src}/{a}", f"{dst}/{tag}_{b}")
out = {}
for name in ("fetch", "write"):
    f = glob.glob(f"{src}/pmc_{name}/**/*counter_collection.csv", recursive=True)[0]
    vals, res = [], None
    for r in csv.DictReader(open(f)):
        if "grx_step_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            res = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
            cname = r["Counter_Name"]
    out[cname] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals)}
    out["kernel_resources"] = res
bench = json.loads(open(f"{src}/bench_rough.json").read())
alg = bench["roofline"]["algorithmic_bytes_per_env_step"] * bench["config"]["envs_per_gpu"]
out["algorithmic_bytes_per_launch"] = alg
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 1000 --warmup 100 "
               "--no-cpu-baseline` (rough terrain, 4096 envs). Units: KB per launch of grx_step_kernel. gfx950 caveat "
               "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads; this kernel's loads "
               "are 4-byte-per-lane SoA columns (uncalibrated width), so the fetched bytes lie between 1x and 2x the counter. "
               "WRITE_SIZE is uncalibrated.")
json.dump(out, open(f"{dst}/{tag}_pmc_hbm_rough4096.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
