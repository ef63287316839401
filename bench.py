#!/usr/bin/env python3
"""bench.py -- constraint-projections/s and ms/substep on the BASELINE.json headline config:
one 1000x1000 cloth (1 000 000 particles; 2 996 001 XPBD distance + 2 992 005 XPBD isometric
bending constraints; 27 colour groups), 10 solver iterations, 1 substep per step, per GPU.

A "step" is one substep of the hot path (integrate -> 10 x colour-ordered projection sweeps ->
velocity update) over the whole sheet, device-resident (inputs already in HBM when the timed
region starts).  With --gpus N every rank owns one independent sheet (ensemble sharding, no
data-path collective; RCCL only for the barrier and the max-over-ranks time): weak scaling,
value = total projections of all ranks / max time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def build_cloth(pbd, n, cloth_method=4, bending_method=3):
    from tests import util
    t0 = time.perf_counter()
    model = util.build_mine(util.cloth_spec(n, n, cloth_method, bending_method))
    model.initConstraintGroups()
    return model, time.perf_counter() - t0


def cpu_baseline(n, iters, budget_s=30.0):
    """The reference's own TimeStepController::step (oracle/_ref, release-like build) timed on the
    host cores of this box on a BOUNDED sample of the same workload: a 400x400 sheet of the same
    cloth (XPBD distance + XPBD isometric bending, same iteration count; projections/s on the CPU
    is size-insensitive), at 1 OpenMP thread and at a few multi-thread
    settings -- the reference forks/joins one parallel region per colour group, which makes large
    thread counts SLOWER, so the best setting is reported as `value`."""
    try:
        from oracle import refdrv
        from tests import util
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "projections/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    variant = "fast" if refdrv.available("fast") else ("f32" if refdrv.available("f32") else None)
    if variant is None:
        from oracle import port
        o = port.Port("f32")
        kind = "port"
        thread_settings = [1]
    else:
        o = refdrv.Ref(variant)
        kind = "reference"
        ncpu = os.cpu_count() or 1
        thread_settings = sorted(set([1, min(8, ncpu), min(32, ncpu)]))
    size = min(n, 400)
    t_setup = time.perf_counter()
    util.apply_ref(o, util.cloth_spec(size, size, 4, 3))
    o.set_time_step_size(0.005)
    o.set_params(1, iters, 0)
    nc = o.num_constraints()
    o.set_num_threads(1)
    o.step(1)   # warm-up: includes the one-off colouring
    t_setup = time.perf_counter() - t_setup
    results = {}
    t_used = 0.0
    for th in thread_settings:
        if t_used > budget_s:
            break
        o.set_num_threads(th)
        o.step(1)
        steps, t = 0, 0.0
        while steps < 2 or (t < 3.0 and steps < 10):
            t += o.time_steps(1)
            steps += 1
            if t > budget_s / len(thread_settings):
                break
        t_used += t
        results[th] = (nc * iters * steps / t, 1e3 * t / steps, steps)
    best = max(results, key=lambda k: results[k][0])
    return {"value": results[best][0], "unit": "projections/s", "cores": best, "kind": kind,
            "ms_per_substep": results[best][1],
            "by_threads": {str(k): {"projections_per_s": v[0], "ms_per_substep": v[1], "timed_steps": v[2]} for k, v in results.items()},
            "sample": "%dx%d cloth (same constraints as the GPU workload, %d constraints), %d iterations x 1 substep, >=2 timed steps per thread setting after warm-up; %s%s; best of OMP threads %s (host reports %d logical CPUs); setup %.1fs" % (
                size, size, nc, iters, "reference build '%s'" % variant if variant else "plain-C port",
                " (-O3 -march=x86-64-v3 -fopenmp, float)" if variant == "fast" else "", sorted(results), os.cpu_count() or 1, t_setup)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=1000, help="cloth is size x size particles")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--xcd-remap", type=int, default=None)
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph")
    ap.add_argument("--fuse", type=int, default=None, help="1 = colour-fused LDS tiles (default), 0 = one launch per colour")
    ap.add_argument("--tile", type=int, default=None, help="particles owned by one tile (0 = auto)")
    ap.add_argument("--fuse-block", type=int, default=None)
    ap.add_argument("--max-seg", type=int, default=None, help="max colours fused into one launch")
    ap.add_argument("--lds-particles", type=int, default=None)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    import positionbaseddynamics_amd as pbd
    if pbd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")

    model, t_build = build_cloth(pbd, args.size)
    n_particles = model.getParticles().size()
    n_constraints = model.numConstraints()
    n_groups = len(model.getConstraintGroups())

    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController(device=local_rank)
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, args.iters)
    sol = ts.solver()
    if args.xcd_remap is not None:
        sol.set_option(pbd.Solver.OPT_XCD_REMAP, args.xcd_remap)
    if args.block is not None:
        sol.set_option(pbd.Solver.OPT_BLOCK_SIZE, args.block)
    if args.no_graph:
        sol.set_option(pbd.Solver.OPT_USE_GRAPH, 0)
    for val, opt in ((args.fuse, pbd.Solver.OPT_FUSE), (args.tile, pbd.Solver.OPT_TILE_PARTICLES), (args.fuse_block, pbd.Solver.OPT_FUSE_BLOCK),
                     (args.max_seg, pbd.Solver.OPT_MAX_SEGMENT_COLOURS), (args.lds_particles, pbd.Solver.OPT_LDS_PARTICLES)):
        if val is not None:
            sol.set_option(opt, val)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed): uploads the device image, instantiates the hipGraph, activates the constraints
    ts.stepResident(model, max(args.warmup, 1))
    barrier()
    t0 = time.perf_counter()
    ts.stepResident(model, args.steps)     # synchronises its own stream before returning
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    stats = sol.stats()
    plan = sol.plan_info()
    barrier()
    if dist is not None:
        tt = torch.tensor([t_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    else:
        t_max = t_local

    projections_per_step = n_constraints * args.iters
    value = projections_per_step * args.steps * world / t_max
    ms_per_step = 1e3 * t_max / args.steps

    # sanity: the state must be finite and the pinned corners must not have moved
    ts.syncToHost(model)
    x = model.getParticles().positions()
    x0 = model.getParticles().array(1)
    ok = bool(np.all(np.isfinite(x)) and np.array_equal(x[0], x0[0]) and np.array_equal(x[args.size - 1], x0[args.size - 1]))

    out = {
        "metric": "constraint-projections/s", "value": value, "unit": "projections/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_substep": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: single %dx%d cloth sheet per GPU (XPBD distance k=1e5 + XPBD isometric bending k=100), %d iterations, 1 substep, h=0.005" % (args.size, args.size, args.iters),
                   "particles": n_particles, "constraints": n_constraints, "colour_groups": n_groups,
                   "projections_per_substep": projections_per_step, "parallelism": "ensemble x%d (one sheet per GPU, no cross-GPU constraints)" % world,
                   "state_ok": ok, "device_event_ms_per_substep": stats["total_ms"] / max(args.steps, 1),
                   "algorithmic_GB_per_substep": stats["algorithmic_bytes"] / max(args.steps, 1) / 1e9,
                   "whole_substep_algorithmic_GBs": stats["algorithmic_bytes"] / max(stats["total_ms"], 1e-9) / 1e6,
                   "host_scene_build_s": t_build, "plan": plan, "engine": sol.describe()},
    }

    if rank == 0 and not args.no_roofline:
        # kernel durations measured live with HIP events on the engine's own stream (eager launches, one
        # event in front of every projection launch of the timed configuration)
        sol.set_profiling(True)
        psteps = max(2, min(5, args.steps))
        ts.stepResident(model, psteps)
        sol.set_profiling(False)
        T = pbd.ConstraintType
        if plan["active"]:
            segs = [sol.segment_info(i) for i in range(plan["num_segments"])]
            rows = []
            for i, si in enumerate(segs):
                if not si["profiled_launches"]:
                    continue
                dur_s = 1e-3 * si["profiled_ms"] / si["profiled_launches"]
                rows.append({"segment": i, "colours": [si["colour_begin"], si["colour_end"]], "tiles": si["num_tiles"], "block": si["block"],
                             "lds_bytes": si["lds_bytes"], "constraints": si["constraints"], "slots": si["slots"],
                             "launches": si["profiled_launches"], "avg_us": dur_s * 1e6,
                             "algorithmic_bytes_per_launch": si["algorithmic_bytes"], "streamed_bytes_per_launch": si["stream_bytes"],
                             "algorithmic_GBs": si["algorithmic_bytes"] / dur_s / 1e9, "streamed_GBs": si["stream_bytes"] / dur_s / 1e9})
            if rows:
                dom = max(rows, key=lambda r: r["avg_us"] * r["launches"])
                out["roofline"] = {"bound": "hbm", "kernel": "fused_kernel (colour-fused LDS tiles), segment %d = colours [%d,%d)" % (dom["segment"], dom["colours"][0], dom["colours"][1]),
                                   "achieved": dom["algorithmic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["algorithmic_GBs"] / HBM_PEAK_GBS,
                                   "traffic": None, "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                                   "streamed_bytes_per_launch": dom["streamed_bytes_per_launch"], "avg_launch_us": dom["avg_us"],
                                   "launches_measured": dom["launches"], "segments": rows,
                                   "note": "achieved = SURVEY 8d algorithmic bytes of the segment's distinct constraints / event-measured launch time; "
                                           "positions stay in LDS for the whole segment, so the bytes actually streamed from HBM (streamed_*) are lower"}
        else:
            per_type = {}
            for t in range(T.COUNT):
                ms, launches, proj = sol.type_stats(t)
                if launches:
                    per_type[T.name(t)] = {"launches": launches, "avg_us": 1e3 * ms / launches, "projections": proj,
                                           "algorithmic_GBs": proj * T.algorithmic_bytes(t) / ms / 1e6}
            ms, launches, proj = sol.type_stats(T.ISOMETRIC_BENDING_XPBD)
            if launches:
                bytes_per_launch = proj * T.algorithmic_bytes(T.ISOMETRIC_BENDING_XPBD) / launches
                dur_s = 1e-3 * ms / launches
                achieved = bytes_per_launch / dur_s / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "project_kernel<ISOMETRIC_BENDING_XPBD>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                   "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": dur_s * 1e6,
                                   "launches_measured": launches, "per_type": per_type}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.iters)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
