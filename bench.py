#!/usr/bin/env python3
"""bench.py -- env-steps/s of the GRx environment step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 500 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 500 --warmup 50

One "step" = one ``env.step()`` of the hot path (clip actions, 10 fused physics sub-steps,
termination, 24 reward terms, masked reset, observations) over one batch of synthetic actions
already resident in HBM.  Workload (named in ``config.workload``): GR1T1 lower-limb on the
ROUGH-TERRAIN curriculum heightfield (10x20 tiles, 1300x2100 int16, seed 1) with the 121-point
height scan, 4096 envs per GPU, all domain randomisation / observation noise / pushes on,
action latency fixed at 5 sub-steps.  Multi-GPU: envs are sharded by global index,
4096 per rank (weak scaling); the step needs NO collective -- the barrier + max-over-ranks below
is measurement only.

Extra objects on the JSON line:
  roofline      HBM roofline of the fused step kernel: algorithmic bytes per env-step
                (2476 B rough / 1750 B flat) x envs per launch / average kernel
                duration measured live with a HIP event pair around the timed region's launches, on the launch stream
                ((stop - start) / steps; GRX_BENCH_EVENT_STRIDE=n: a pair around every n-th launch, grx_kernel_time_ms).
                `kernel` / `layout`: what grx_layout() says the handle launches (not guessed from the batch size).
                `valu_issue_frac`: the number that actually bounds this kernel -- VALU wave-instructions
                per launch (SQ_INSTS_VALU of the committed rocprofv3 PMC pass of the same workload and batch size,
                profiles/rNN_pmc_sq_<workload>.json, newest round first)
                / live kernel duration, against the chip's VALU issue peak (1024 SIMDs x one wave64
                instruction per 2 cycles at 2.4 GHz, MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle ("port": oracle/grx_oracle.c, fp32, OpenMP over envs) timed on
                this box's host cores on a bounded sample of the same workload.  It is NOT
                Isaac Gym's CPU pipeline (unobtainable).  `reference_stage`
                quotes the one piece of the reference that CAN run: its own torch-CPU
                post_physics_step (observations / rewards / resets, no physics), timed in the build
                container (tools/time_ref_pipeline.py -> profiles/r02_ref_python_pipeline.json).
"""
import argparse
import os

import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_FLAT, B_ROUGH = 1750.0, 2476.0   # algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0   # wave64 VALU instructions / s: 256 CUs x 4 SIMDs, one per 2 cycles, 2.4 GHz


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_stage():
    """The reference's own CPU number for the ONE stage of the path it can run without Isaac Gym (build-container
    measurement, committed): read from profiles/, never measured here (the reference does not travel)."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r02_ref_python_pipeline.json")))
        row = [r for r in j["rows"] if r["num_envs"] == 4096 and r["terrain"].startswith("rough")][0]
        return {"value": row["env_steps_per_s_stage_only"], "unit": "env-steps/s", "cores": j["cores"], "cpu_model": j["cpu_model"],
                "what": "reference post_physics_step (torch CPU: observations, 34 reward terms, resets; NO physics), 4096 envs rough, build container",
                "source": "profiles/r02_ref_python_pipeline.json"}
    except Exception:
        return None


def make_cfg(terrain, robot="lower_limb"):
    from wiki_grx_gym_amd.envs import config
    # registered task "GR1T1" = lower-limb config (the headline); --robot full_body = BASELINE.json config 5 (32 DOF)
    # --robot gr1t2 = the robot of BASELINE.json's fourth configuration (GR1T2 lower limb, same fused kernels; a rank's 4096-env shard)
    cfg = config.GR1T1Cfg() if robot == "lower_limb" else config.GR1T2Cfg() if robot == "gr1t2" else config.GR1T1FullBodyCfg()
    cfg.terrain.mesh_type = {"rough": "heightfield", "trimesh": "trimesh", "flat": "plane"}[terrain]
    cfg.terrain.curriculum = True
    # THE PRODUCT DEFAULT (round 5): the tensors nobody reads in a rollout -- rigid_body_states ("not counted, 1924 B"),
    # measured_heights -- are published ON REFRESH (include/grx.h grx_publish_mode: materialised by grx_refresh when read), which is what
    # task_registry.make_env("GR1T1") builds too.  GRX_BENCH_RBS=1: both written by every step instead (the surcharge measurement).
    if os.environ.get("GRX_BENCH_RBS", "0") == "1":
        cfg.env.publish_rigid_body_states = cfg.env.publish_measured_heights = "every_step"
    return cfg


def full_iteration(args, n_local, iters, warm=2):
    """The reference's OWN throughput metric (rsl_rl/runners/on_policy_runner.py:235, 242: fps = num_steps_per_env * num_envs /
    (collection_time + learn_time)): whole PPO iterations -- 64 policy steps of rollout with the actor / critic in the loop, GAE,
    8 epochs x 25 minibatches of the update -- on the env exactly as task_registry.make_env builds it (the product default), PPO
    hyper-parameters of the registered task.  `warm` untimed iterations first (HIP-graph capture of the update and of the policy step)."""
    import torch
    import torch.distributed as dist
    from wiki_grx_gym_amd.envs import config
    from wiki_grx_gym_amd.utils import get_args, task_registry
    task = {"lower_limb": "GR1T1", "gr1t2": "GR1T2", "full_body": "GR1T1_full_body"}[args.robot]
    a = get_args(["--task", task, "--headless", "--num_envs", str(n_local), "--seed", "1"])
    cfg = make_cfg(args.terrain, args.robot)
    cfg.seed = 1
    env, _ = task_registry.make_env(task, args=a, env_cfg=cfg)
    tcfg = config.GR1T1CfgPPO() if args.robot == "lower_limb" else config.GR1T2CfgPPO() if args.robot == "gr1t2" else config.GR1T1FullBodyCfgPPO()
    tcfg.seed = 1
    runner, _ = task_registry.make_alg_runner(env, name=task, args=a, train_cfg=tcfg, log_root=None)
    runner.sync_timers = True
    runner.learn(warm, init_at_random_ep_len=True)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    coll = learn = 0.0
    for _ in range(iters):
        runner.learn(1)
        coll += runner.last_collection_time; learn += runner.last_learn_time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    T = runner.num_steps_per_env
    out = {"env_steps_per_s": T * n_local * world * iters / dt, "unit": "env-steps/s (training: rollout + PPO update)", "iters": iters, "warm_iters": warm,
           "collection_ms": coll / iters * 1e3, "learn_ms": learn / iters * 1e3, "iteration_ms": dt / iters * 1e3,
           "num_steps_per_env": T, "envs_per_gpu": n_local, "n_gpus": world,
           "rollout_env_steps_per_s": T * n_local * world / (coll / iters) if coll > 0 else None,
           "definition": "num_steps_per_env * num_envs / (collection_time + learn_time), rsl_rl/runners/on_policy_runner.py:235 -- over whole iterations by the wall clock "
                         "(max over ranks), device drained at both ends; collection_ms / learn_ms: rank 0's split with the device drained at the boundary",
           "policy": "ActorCritic MLP [512, 256, 128] actor + critic, PPO 8 epochs x 25 minibatches, adaptive LR (the registered task's " + type(tcfg).__name__ + ")",
           "env": "task_registry.make_env default (rigid_body_states / measured_heights on refresh), action latency drawn per step N(5, 2) as in training",
           "multi_gpu": None if world == 1 else "envs sharded by global index; ONE flat-bucket RCCL all-reduce of the gradients per optimizer step (DESIGN.md 7)"}
    env.close()
    return out


def cpu_baseline(cfg, terrain_obj, envs, steps, seed):
    """Oracle (fp32, OpenMP) on the host cores: bounded sample of the same workload."""
    import torch
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd.envs import build_config
    from tests.helpers import random_actions
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, envs, 0, envs, seed, terrain_obj)
    sim = OracleSim(c, "f32", keep)
    sim.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, envs, gen, 1.0) for _ in range(4)]
    sim.step(acts[0], 5.0, 1)
    t0 = time.time()
    for i in range(steps):
        sim.step(acts[i % 4], 5.0, 2 + i)
    dt = time.time() - t0
    sim.close()
    return {"value": envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{envs} envs x {steps} steps of the same workload, oracle/grx_oracle.c fp32 + OpenMP ({dt:.1f} s)",
            "note": "a C restatement of THIS build's algorithm (physics + env pipeline) on the host cores -- not the reference: Isaac Gym's "
                    "CPU pipeline is a closed binary absent from this image (BASELINE.md 2-3, B-cpu).  The reference's own runnable CPU "
                    "stage is quoted in reference_stage.",
            "reference_stage": reference_stage()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed window of seconds, not milliseconds.  On the (shared) MI355X boxes the first ~100 ms of a
    # process's launches regularly contain one 10-80 ms hole in which the GPU runs nothing of this job (DESIGN.md
    # section 5); 500 steps are 39 ms of work, so most default runs used to report that hole instead of the kernel.
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20000; 2000 for --robot full_body)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: steps / 10)")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--terrain", choices=["rough", "flat", "trimesh"], default="rough",
                    help="rough: the curriculum raster as a heightfield (BASELINE.json's configurations); trimesh: the same raster as the reference's slope-corrected triangle mesh (vertical faces)")
    ap.add_argument("--robot", choices=["lower_limb", "gr1t2", "full_body"], default="lower_limb",
                    help="full_body: the 32-DOF GR1T1 of BASELINE.json config 5 (tree kernel, grx_tree.h), not the headline; "
                         "gr1t2: the GR1T2 lower-limb robot of config 4 on the headline's kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-iters", type=int, default=5, help="PPO iterations timed for the `full_iteration` object (the reference's own fps metric, "
                    "on_policy_runner.py:235); 0 = skip it")
    ap.add_argument("--train-timeout", type=float, default=150.0, help="seconds after which the full_iteration leg is given up (the line is printed without it)")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=0, help="0 = auto (about 15 s of CPU work)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2000 if args.robot == "full_body" else 20000
    if args.warmup is None:
        args.warmup = max(1, args.steps // 10)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GRx step has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # launched by torch.distributed.run (RANK set): one process per GPU over RCCL -- the process group, the barriers and the
    # max-over-ranks reduction below run at every world size, 1 included (tests/test_env_gpu.py drives that on one GPU)
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from wiki_grx_gym_amd.envs import build_config
    from wiki_grx_gym_amd.sim import HipSim
    from wiki_grx_gym_amd.utils.terrain import Terrain
    from tests.helpers import random_actions

    seed = 1
    n_local = args.envs_per_gpu
    n_total = n_local * world
    cfg = make_cfg(args.terrain, args.robot)
    terrain_obj = Terrain(cfg.terrain, n_total, seed=seed) if args.terrain != "flat" else None
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, n_local, rank * n_local, n_total, seed, terrain_obj)
    os.environ.setdefault("GRX_PUBLISH_DEBUG", "0")   # production path: no per-term debug tensors
    sim = HipSim(c, dev, keep)
    sim.reset_all()
    gen = torch.Generator().manual_seed(rank)
    pool = [random_actions(cfg, n_local, gen, 0.3 if args.robot == "full_body" else 1.0).to(dev) for _ in range(16)]   # U[clip_min, clip_max]
    delay = 5.0
    counter = 0
    # The GPU idles while the host builds the terrain and the robot tables (seconds): its first kernels then run at the idle power
    # state's clocks.  Measured on the driver's own window (--steps 20 --warmup 5, round 4, three runs each): no pre-spin 76.6-78.5 M
    # env-steps/s, 250 ms of a plain matmul 80.6-82.3 M, 250 ms of THIS workload on a second, throw-away handle 82.3-84.1 M (the
    # kernel's own power / clock regime; the 20000-step default reads 88 M either way).  So the device is kept busy for `prespin_ms`
    # BEFORE the warm-up steps with steps of a second handle (GRX_BENCH_PRESPIN_KIND=matmul: the matmul of rounds 2-3); no step of the
    # timed handle is skipped, added or moved, and kind and duration are in the JSON line (GRX_BENCH_PRESPIN_MS=0 turns it off).
    prespin = float(os.environ.get("GRX_BENCH_PRESPIN_MS", "250"))
    prespin_kind = os.environ.get("GRX_BENCH_PRESPIN_KIND", "step")
    if prespin > 0 and prespin_kind == "step":   # a SECOND handle of the same workload keeps the device busy: this kernel's own power / clock regime
        c2, keep2, _ = build_config.build(cfg, cfg.sim.dt, n_local, rank * n_local, n_total, seed + 1, terrain_obj)
        sim2 = HipSim(c2, dev, keep2)
        sim2.reset_all()
        t_ = time.perf_counter(); k_ = 0
        while (time.perf_counter() - t_) * 1e3 < prespin:
            for _ in range(64):
                k_ += 1; sim2.step(pool[k_ % 16], delay, k_)
            sim2.wait_idle()
        sim2.close()
    elif prespin > 0:
        a_ = torch.randn(4096, 4096, device=dev)
        t_ = time.perf_counter()
        while (time.perf_counter() - t_) * 1e3 < prespin:
            a_ = (a_ @ a_) * 1e-4
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        counter += 1
        sim.step(pool[counter % 16], delay, counter)
    # The step kernel's launch duration, live, by HIP events on the stream it is launched on (HipSim launches on torch's current stream):
    # ONE pair around the K launches of the timed region -- (stop - start) / K, the gaps between back-to-back launches included, so an
    # upper bound of the kernel's own duration (rocprofv3 reads 0.7 us less per launch at 4096 envs).  GRX_BENCH_EVENT_STRIDE=n > 0
    # brackets every n-th launch with a pair of its own inside the library instead (rounds 1-4: a pair costs the stream ~7 us and brackets
    # the dispatch as well -- 48.8 us where rocprofv3 reads 45.8 -- and three of them are 2 % of the driver's 20-step window).
    ev_stride = int(os.environ.get("GRX_BENCH_EVENT_STRIDE", "0"))
    if ev_stride > 0:
        sim.kernel_time_ms(enable=ev_stride)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if ev_stride <= 0:
        ev0.record()   # (the device is idle: stamped at once; enqueued ahead of the host clock's start so that its ~5 us call is not in `value`)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counter += 1
        sim.step(pool[counter % 16], delay, counter)
    if ev_stride <= 0:
        ev1.record()
    sim.wait_idle()           # spin on the library's pinned progress word until the last step has finished: the HIP
    torch.cuda.synchronize()  # runtime's own completion view was measured to lag by 10-80 ms sporadically
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()   # (one process: the synchronize above is the bracket; a second one on an idle device is ~10 us of a 1 ms window)
    elapsed = time.perf_counter() - t0
    if ev_stride > 0:
        kern_ms, launches = sim.kernel_time_ms(enable=False)
    else:
        kern_ms, launches = ev0.elapsed_time(ev1) / args.steps, args.steps
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(sim.tensor("OBS")).all().item()) and bool(torch.isfinite(sim.tensor("REW")).all().item())
    out = None
    layout = sim.layout()
    if rank == 0:
        bytes_per = B_ROUGH if args.terrain != "flat" else B_FLAT
        if args.robot == "full_body":   # B_full = 3422 B/env-step (+726 of height gathers on rough terrain)
            bytes_per = 3422.0 + (726.0 if args.terrain != "flat" else 0.0)
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this very workload (separate FETCH_SIZE /
        # WRITE_SIZE runs, tools/collect_profiles.sh); bench.py cannot host the profiler itself.  Raw counter sum:
        # FETCH_SIZE is a lower bound on gfx950 (MI355X_MICROARCH.md), see the note inside the file.
        traffic, traffic_src, valu_insts, valu_src = None, None, None, None
        wl = {"lower_limb": "", "gr1t2": "gr1t2_", "full_body": "full_body_"}[args.robot] + f"{args.terrain}{n_local}"   # e.g. rough4096, full_body_rough4096
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_{wl}.json")
            if traffic is None and os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    traffic = (j["FETCH_SIZE"]["mean_KB"] + j["WRITE_SIZE"]["mean_KB"]) * 1024.0
                    if j.get("kernel") and not layout["kernel"].startswith(j["kernel"].split("<")[0] + "<"):
                        raise ValueError("counters of another kernel")
                    traffic_src = (f"profiles/{tag}_pmc_hbm_{wl}.json: FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this "
                                   "workload, bytes per launch, NOT live and uncorrected (FETCH_SIZE is a lower bound on gfx950)")
                except Exception:
                    traffic = None
            sq = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_{wl}.json")
            if valu_insts is None and os.path.exists(sq):
                try:
                    valu_insts = float(json.load(open(sq))["SQ_INSTS_VALU"])
                    valu_src = f"profiles/{tag}_pmc_sq_{wl}.json (SQ_INSTS_VALU per launch, rocprofv3 --pmc pass of this workload; duration live)"
                except Exception:
                    valu_insts = None
        achieved = bytes_per * n_local / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": f"env-steps/sec {'GR1T2' if args.robot == 'gr1t2' else 'GR1T1'} {'flat' if args.terrain == 'flat' else 'rough'}-terrain{' (trimesh)' if args.terrain == 'trimesh' else ''} @{n_local} envs"
                      + (" [full-body 32 DOF, config 5]" if args.robot == "full_body" else ""),
            "value": n_total * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{'GR1T2' if args.robot == 'gr1t2' else 'GR1T1'} {'full body (32 DOF, tree kernel)' if args.robot == 'full_body' else 'lower-limb (10 DOF)'}, {'flat plane' if args.terrain == 'flat' else 'rough-terrain curriculum ' + ('trimesh (the raster as the slope-corrected triangle mesh, vertical faces)' if args.terrain == 'trimesh' else 'heightfield') + ' 10x20 tiles + 121-pt height scan'}, "
                                   f"{n_local} envs/GPU, decimation 10 @ dt 0.002, DR+noise+push on, action latency 5 sub-steps, random actions U[clip_min,clip_max]",
                       "envs_per_gpu": n_local, "global_envs": n_total, "parallelism": f"env-sharded x{world} (no data-path collective)",
                       "finite_outputs": finite, "prespin_ms": prespin, "prespin_kind": prespin_kind,
                       # grx_publish_mode of rigid_body_states / measured_heights: "on_refresh" = the product default (what make_env builds);
                       # GRX_BENCH_RBS=1 = "every_step" (both written by the step kernel: the surcharge measurement, +2.4 KB per env-step)
                       "on_demand_tensors": str(getattr(cfg.env, "publish_rigid_body_states", True) is True and "on_refresh" or getattr(cfg.env, "publish_rigid_body_states", True)),
                       "product_default": os.environ.get("GRX_BENCH_RBS", "0") != "1",
                       "layout": layout},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per * n_local,
                         "kernel": layout["kernel"], "kernel_ms": kern_ms, "launches_timed": launches,
                         "kernel_ms_method": ("one HIP event pair around the timed region's launches, (stop - start) / steps: the gaps between launches included" if ev_stride <= 0
                                              else f"HIP event pairs around every {ev_stride}th launch (each pair brackets the dispatch as well)"),
                         "algorithmic_bytes_per_env_step": bytes_per,
                         "valu_issue_frac": (valu_insts / (kern_ms * 1e-3) / VALU_ISSUE_PEAK) if (valu_insts and kern_ms > 0) else None,
                         "valu_insts_per_launch": valu_insts, "valu_issue_peak_per_s": VALU_ISSUE_PEAK, "valu_source": valu_src,
                         "note": (f"instruction-issue bound at this batch size ({layout['lanes_per_env']} lanes per env, {layout['waves_per_block']} waves per {layout['envs_per_block']}-env block, "
                                  f"{layout['num_blocks']} blocks: DESIGN.md sections 4.1 and 5); HBM is the contractual roofline"
                                  if args.robot != "full_body" else "tree kernel (a lane group per env, a chain per lane): instruction-issue bound (DESIGN.md section 4.3); HBM is the contractual roofline")},
        }
    sim.close()
    # ---- the reference's own metric beside the headline (VERDICT r4 missing #2): whole PPO iterations, every rank takes part.  Guarded by a
    # watchdog: were the training leg ever to hang (a collective on a node this build has never run on), rank 0 still prints the headline.
    full_it = None
    if args.train_iters > 0:
        import threading
        done = threading.Event()

        def give_up():
            if done.is_set():
                return
            if rank == 0:
                out["full_iteration"] = {"error": f"gave up after {args.train_timeout:.0f} s"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        wd = threading.Timer(args.train_timeout, give_up)
        wd.daemon = True
        wd.start()
        try:
            full_it = full_iteration(args, n_local, args.train_iters)
        except Exception as ex:   # the headline does not depend on the training leg
            full_it = {"error": f"{type(ex).__name__}: {ex}"[:400]}
        finally:
            done.set(); wd.cancel()
    if rank == 0:
        if full_it is not None:
            out["full_iteration"] = full_it
        if world == 1 and not args.no_cpu_baseline:
            steps = args.cpu_steps
            if steps <= 0:   # calibrate on a short probe, then aim at ~15 s (bounded to [3, 400] steps)
                probe = cpu_baseline(cfg, terrain_obj, args.cpu_envs, 4, seed)
                steps = min(400, max(3, int(15.0 * probe["value"] / args.cpu_envs)))
            out["cpu_baseline"] = cpu_baseline(cfg, terrain_obj, args.cpu_envs, steps, seed)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
