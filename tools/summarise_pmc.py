"""profiles/<tag>_pmc_hbm_<workload>.json and profiles/<tag>_pmc_sq_<workload>.json from the passes tools/collect_pmc.sh left under
gpurun_out/<tag>/pmc/<workload>/ (mean per launch of the step kernel; what bench.py's roofline.traffic / valu_issue_frac read).
usage: python tools/summarise_pmc.py <tag> <workload> [algorithmic bytes per launch]"""
import csv, glob, json, os, sys
tag, wl = sys.argv[1], sys.argv[2]
src = f"gpurun_out/{tag}/pmc/{wl}"
cmd = open(f"{src}/command.txt").read().strip() if os.path.exists(f"{src}/command.txt") else ""
STEP = ("grx_step_kernel", "grx_step_tree", "grx_step_generic")


def rows(name):
    fs = glob.glob(f"{src}/{name}/**/*counter_collection.csv", recursive=True)
    return [r for r in csv.DictReader(open(fs[0])) if any(k in r["Kernel_Name"] for k in STEP)] if fs else []


def kname(r):
    return r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


hbm = {}
for name in ("fetch", "write"):
    rs = rows(name)
    if rs:
        v = [float(r["Counter_Value"]) for r in rs]
        hbm[rs[0]["Counter_Name"]] = {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
        hbm["kernel"] = kname(rs[0])
        hbm["kernel_resources"] = {k: rs[0][k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in rs[0]}
if hbm:
    if len(sys.argv) > 3:
        hbm["algorithmic_bytes_per_launch"] = float(sys.argv[3])
        hbm["traffic_over_algorithmic"] = (hbm["FETCH_SIZE"]["mean_KB"] + hbm["WRITE_SIZE"]["mean_KB"]) * 1024 / float(sys.argv[3])
    hbm["note"] = (f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `{cmd}`; KB per launch of the step kernel, uncorrected. gfx950 caveat "
                   "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads; this kernel's loads are 4-byte-per-lane SoA columns, "
                   "so the fetched bytes lie between 1x and 2x the counter. WRITE_SIZE is uncalibrated.")
    json.dump(hbm, open(f"profiles/{tag}_pmc_hbm_{wl}.json", "w"), indent=1)
sq = {"workload": f"{cmd}; rocprofv3 --pmc, one pass per counter group (tools/collect_pmc.sh), mean per launch of the step kernel"}
for name in ("sq1", "sq2", "sq3", "lds"):
    agg = {}
    for r in rows(name):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        sq["kernel"] = kname(r)
    for k, v in agg.items():
        sq[k] = round(sum(v) / len(v))
if "SQ_INSTS_VALU" in sq:
    sq["derived"] = {"valu_per_wave": sq["SQ_INSTS_VALU"] / max(sq.get("SQ_WAVES", 1), 1), "lds_per_wave": sq.get("SQ_INSTS_LDS", 0) / max(sq.get("SQ_WAVES", 1), 1),
                     "wait_any_fraction": sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
                     "lds_bank_conflict_fraction": sq.get("SQ_LDS_BANK_CONFLICT", 0) / max(sq.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                     "note": "SQ_WAIT_ANY includes the helper waves' spin on the LDS sequence flags; lds_bank_conflict_fraction = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"}
    json.dump(sq, open(f"profiles/{tag}_pmc_sq_{wl}.json", "w"), indent=1)
print(wl, hbm.get("kernel"), {k: round(v["mean_KB"]) for k, v in hbm.items() if isinstance(v, dict) and "mean_KB" in v}, {k: sq[k] for k in ("SQ_INSTS_VALU", "SQ_WAVES") if k in sq}, sq.get("derived", {}).get("lds_bank_conflict_fraction"))
