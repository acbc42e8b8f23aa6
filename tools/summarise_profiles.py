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
    out["kernel"] = (res or {}).get("Kernel_Name", "").replace("void ", "").split("(")[0]
bench = json.loads(open(f"{src}/bench_rough.json").read())
alg = bench["roofline"]["algorithmic_bytes_per_env_step"] * bench["config"]["envs_per_gpu"]
out["algorithmic_bytes_per_launch"] = alg
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 1000 --warmup 100 "
               "--no-cpu-baseline` (rough terrain, 4096 envs). Units: KB per launch of grx_step_kernel. gfx950 caveat "
               "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads; this kernel's loads "
               "are 4-byte-per-lane SoA columns (uncalibrated width), so the fetched bytes lie between 1x and 2x the counter. "
               "WRITE_SIZE is uncalibrated.")
json.dump(out, open(f"{dst}/{tag}_pmc_hbm_rough4096.json", "w"), indent=1)
# SQ passes -> one JSON (mean per launch of grx_step_kernel)
sq = {"kernel": "grx_step_kernel_quad<true, 8, false> (the headline layout at 4096 envs: eight waves per 16-env block)", "workload": "python bench.py --steps 300 --warmup 50 --no-cpu-baseline (rough, 4096 envs); rocprofv3 --pmc, "
      "one pass per counter group (tools/collect_profiles.sh), mean per launch"}
for f in sorted(glob.glob(f"{src}/pmc_sq*/**/*counter_collection.csv", recursive=True)):
    agg = {}
    for r in csv.DictReader(open(f)):
        if "grx_step_kernel" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k_, v_ in agg.items():
        sq[k_] = round(sum(v_) / len(v_))
if "SQ_INSTS_VALU" in sq:
    kms = bench["roofline"]["kernel_ms"]
    sq["derived"] = {"kernel_ms_of_the_bench_line": kms,
                     "valu_issue_frac": sq["SQ_INSTS_VALU"] / (kms * 1e-3) / (256 * 4 * 2.4e9 / 2.0),
                     "valu_per_wave": sq["SQ_INSTS_VALU"] / max(sq.get("SQ_WAVES", 1), 1),
                     "wait_any_fraction": sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
                     "note": "VALU issue peak = 1024 SIMDs x one wave64 instruction per 2 cycles x 2.4 GHz; SQ_WAIT_ANY includes the helper waves' "
                             "spin on the LDS sequence flags"}
    json.dump(sq, open(f"{dst}/{tag}_pmc_sq_rough4096.json", "w"), indent=1)
if os.path.exists(f"{src}/layouts.jsonl"):
    shutil.copy(f"{src}/layouts.jsonl", f"{dst}/{tag}_bench_n1_rough4096_layouts.jsonl")
for n in (4096, 16384):
    if os.path.exists(f"{src}/bench_full_body_rough{n}.json"):
        shutil.copy(f"{src}/bench_full_body_rough{n}.json", f"{dst}/{tag}_bench_n1_full_body_rough{n}.json")
fb = glob.glob(f"{src}/stats_full_body/**/*kernel_stats.csv", recursive=True)
if fb:
    shutil.copy(fb[0], f"{dst}/{tag}_kernel_stats_full_body_rough16384.csv")
fb = glob.glob(f"{src}/stats_full_body4096/**/*kernel_stats.csv", recursive=True)
if fb:
    shutil.copy(fb[0], f"{dst}/{tag}_kernel_stats_full_body_rough4096.csv")
fb = glob.glob(f"{src}/stats_8192/**/*kernel_stats.csv", recursive=True)
if fb:
    shutil.copy(fb[0], f"{dst}/{tag}_kernel_stats_rough8192.csv")
    shutil.copy(f"{src}/bench_rough8192.json", f"{dst}/{tag}_bench_n1_rough8192.json")
# config 5 (tree kernel) counter passes
fbo = {}
for name in ("fetch", "write"):
    fs = glob.glob(f"{src}/fb_pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    rows_ = [r for r in csv.DictReader(open(fs[0])) if "grx_step_tree" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows_]
    if rows_:   # the kernel that ran (grx_step_tree<...> with 8 lanes per env, grx_step_tree16<...> with 16): bench.py matches it against grx_layout
        fb_kernel = rows_[0]["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if vals:
        fbo[name.upper() + "_SIZE"] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals)}
if fbo:
    fbo["kernel"] = fb_kernel
    fbo["algorithmic_bytes_per_launch"] = (3422.0 + 726.0) * 4096
    fbo["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `python bench.py --robot full_body --envs-per-gpu 4096 --steps 200`; KB per launch of grx_step_tree; same gfx950 caveats as the lower-limb file"
    json.dump(fbo, open(f"{dst}/{tag}_pmc_hbm_full_body_rough4096.json", "w"), indent=1)
fsq = {"kernel": fb_kernel if "fb_kernel" in dir() else "grx_step_tree<true>", "workload": "python bench.py --robot full_body --envs-per-gpu 4096 --steps 200 --warmup 20 --no-cpu-baseline; rocprofv3 --pmc, one pass per counter group, mean per launch"}
for f in sorted(glob.glob(f"{src}/fb_pmc_sq*/**/*counter_collection.csv", recursive=True)):
    agg = {}
    for r in csv.DictReader(open(f)):
        if "grx_step_tree" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k_, v_ in agg.items():
        fsq[k_] = round(sum(v_) / len(v_))
if "SQ_INSTS_VALU" in fsq:
    fsq["derived"] = {"valu_per_wave": fsq["SQ_INSTS_VALU"] / max(fsq.get("SQ_WAVES", 1), 1), "lds_per_wave": fsq.get("SQ_INSTS_LDS", 0) / max(fsq.get("SQ_WAVES", 1), 1),
                      "wait_any_fraction": fsq.get("SQ_WAIT_ANY", 0) / max(fsq.get("SQ_WAVE_CYCLES", 1), 1)}
    json.dump(fsq, open(f"{dst}/{tag}_pmc_sq_full_body_rough4096.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
