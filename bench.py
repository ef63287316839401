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


def build_workload(args, ens):
    """Returns (model, build seconds, workload description, check function).  Workloads:
       c2  (default, the headline config) one size x size cloth per GPU
       c4  BASELINE configs[3]: `--instances` independent size x size cloths (512 x 200x200 over 8 GPUs
           = 64 per GPU); the global instance list is sharded contiguously over the ranks
       c3  BASELINE configs[2]: 101x21x11 FEM-tet bar (100 000 tets), one per GPU -- latency-bound by
           construction (40 colours x 10 iterations over 23 331 particles), reported, not the headline"""
    from tests import util
    t0 = time.perf_counter()
    if args.workload == "c3":
        k = args.instances if args.bars else 1
        spec = util.bar_spec(101, 21, 11, args.solid_method, instances=k)
        desc = "configs[2]: %s101x21x11 regular tet bar (100000 tets), solid method %d, %d iterations, 1 substep, h=0.005" % (
            ("%d independent bars, each a " % k) if k > 1 else "", args.solid_method, args.iters)
        pins = [0]
    elif args.workload == "c4":
        begin, end = ens.shard(args.instances * ens.world)     # weak scaling: `instances` per GPU
        k = end - begin
        spec = util.cloth_spec(args.size, args.size, 4, 3, instances=k, instance_offset=(0.0, 0.0, 12.0))
        desc = "configs[3]: %d independent %dx%d cloth instances per GPU (XPBD distance + XPBD isometric bending), %d iterations, 1 substep, h=0.005" % (
            k, args.size, args.size, args.iters)
        pins = [0, args.size - 1]
    else:
        spec = util.cloth_spec(args.size, args.size, 4, 3)
        desc = "configs[1]: single %dx%d cloth sheet per GPU (XPBD distance k=1e5 + XPBD isometric bending k=100), %d iterations, 1 substep, h=0.005" % (
            args.size, args.size, args.iters)
        pins = [0, args.size - 1]
    model = util.build_mine(spec)
    model.initConstraintGroups()
    return model, time.perf_counter() - t0, desc, pins


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


CALIB_BYTES = 1 << 28


def _pmc_pass(counter, child_args, timeout_s=240):
    """One `rocprofv3 --pmc <counter>` pass over a short child run of this script (its own process:
    counters and traces are never combined, and the profiled run is never the timed one).
    Returns {kernel_name: [counter values per dispatch]} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    outdir = tempfile.mkdtemp(prefix="pbdx_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
    except Exception:
        return None
    vals = {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    shutil.rmtree(outdir, ignore_errors=True)
    return vals or None


def rocprof_kernel_durations(child_args, kernel_substring, timeout_s=240):
    """Cross-check of the event-measured launch durations: a `rocprofv3 --kernel-trace` child pass (no
    counters) of a short run of the same configuration; returns (mean ns, dispatches) of the dominant kernel."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    outdir = tempfile.mkdtemp(prefix="pbdx_trace_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "trace", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
    except Exception:
        return None
    durs = []
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substring in row.get("Kernel_Name", ""):
                    durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    shutil.rmtree(outdir, ignore_errors=True)
    if not durs:
        return None
    return sum(durs) / len(durs), len(durs)


def collect_traffic(child_args, kernel_substring):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md (HBM)
    prescribes: FETCH_SIZE and WRITE_SIZE in separate passes; the counters are turned into bytes with
    factors calibrated IN THE SAME PASS on streaming kernels of known size in this engine's own access
    widths (the guide: gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read; other widths
    and WRITE_SIZE must be calibrated).  Infinity-Cache hits are counted as traffic."""
    res = {}
    for counter, calib in (("FETCH_SIZE", ("calib_read_b32", "calib_read_b128")), ("WRITE_SIZE", ("calib_write_b32", "calib_write_b128"))):
        vals = _pmc_pass(counter, child_args)
        if not vals:
            return None
        factors = {}
        for name in calib:
            v = [x for k, xs in vals.items() if name in k for x in xs]
            if v:
                factors[name] = CALIB_BYTES / (sum(v) / len(v))      # bytes per counter unit
        dom = [x for k, xs in vals.items() if kernel_substring in k for x in xs]
        if not dom or not factors:
            return None
        res[counter] = {"mean_counter_per_launch": sum(dom) / len(dom), "launches": len(dom), "bytes_per_unit": factors}
    # the dominant kernel reads 4-byte streams (parameters, multipliers, packed indices) and 16-byte
    # positions, writes 4-byte multipliers and 16-byte positions: bracket with both calibrations
    out = {"fetch_bytes": {}, "write_bytes": {}}
    for name, f in res["FETCH_SIZE"]["bytes_per_unit"].items():
        out["fetch_bytes"][name] = res["FETCH_SIZE"]["mean_counter_per_launch"] * f
    for name, f in res["WRITE_SIZE"]["bytes_per_unit"].items():
        out["write_bytes"][name] = res["WRITE_SIZE"]["mean_counter_per_launch"] * f
    fb = out["fetch_bytes"].get("calib_read_b32", next(iter(out["fetch_bytes"].values())))
    wb = out["write_bytes"].get("calib_write_b32", next(iter(out["write_bytes"].values())))
    out["bytes_per_launch"] = fb + wb
    out["raw"] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2")
    ap.add_argument("--size", type=int, default=None, help="cloth is size x size particles (default 1000; 200 for c4)")
    ap.add_argument("--instances", type=int, default=64, help="c4: cloth instances per GPU")
    ap.add_argument("--bars", action="store_true", help="c3: batch --instances independent bars per GPU (the single bar is latency-bound by construction)")
    ap.add_argument("--solid-method", type=int, default=2, help="c3: addSolidConstraints method (2 FEM tet, 4 strain tet, 6 XPBD distance+volume)")
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
    ap.add_argument("--persistent", type=int, default=None, help="PBDX_OPT_PERSISTENT: 1 = one launch per substep where measured faster (default), 0 = one launch per segment, 2 = always")
    ap.add_argument("--contacts", action="store_true", help="also time the step with two static colliders (contact detection + velocity solve per step)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.size is None:
        args.size = 200 if args.workload == "c4" else 1000
    import torch
    from positionbaseddynamics_amd.ensemble import Ensemble
    ens = Ensemble()
    rank, local_rank, world, dist = ens.rank, ens.local_rank, ens.world, ens.dist
    if world == 1:
        torch.cuda.set_device(0)
        local_rank = 0

    import positionbaseddynamics_amd as pbd
    if pbd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")

    if args.pmc_child:
        from positionbaseddynamics_amd import _ffi
        for mode in range(4):
            _ffi.check(_ffi.lib.pbdx_debug_stream(local_rank, CALIB_BYTES, mode), "pbdx_debug_stream")
        args.no_roofline = args.no_cpu_baseline = True
        args.steps, args.warmup = 2, 1

    model, t_build, workload_desc, pins = build_workload(args, ens)
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
                     (args.max_seg, pbd.Solver.OPT_MAX_SEGMENT_COLOURS), (args.lds_particles, pbd.Solver.OPT_LDS_PARTICLES),
                     (args.persistent, pbd.Solver.OPT_PERSISTENT)):
        if val is not None:
            sol.set_option(opt, val)

    def barrier():
        torch.cuda.synchronize()
        ens.barrier()

    # warm-up (untimed): uploads the device image, instantiates the hipGraph, activates the constraints
    ts.stepResident(model, max(args.warmup, 1))
    barrier()
    t0 = time.perf_counter()
    ts.stepResident(model, args.steps)     # synchronises its own stream before returning
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    stats = sol.stats()
    plan = sol.plan_info()
    persist = sol.persistent_info()
    barrier()
    t_max = ens.max_time(t_local)                       # max over ranks
    total_constraints = ens.sum_count(n_constraints)    # all ranks (weak scaling: every rank owns its own instances)

    projections_per_step = n_constraints * args.iters
    value = total_constraints * args.iters * args.steps / t_max
    ms_per_step = 1e3 * t_max / args.steps

    # sanity: the state must be finite and the pinned corners must not have moved
    ts.syncToHost(model)
    x = model.getParticles().positions()
    x0 = model.getParticles().array(1)
    ok = bool(np.all(np.isfinite(x)) and all(np.array_equal(x[p], x0[p]) for p in pins))
    # cross-GPU parity: every rank simulated the same synthetic scene, so every rank must hold the same bits
    # (one tiny all-reduce of per-rank checksums; the only other collectives are the barrier and the max time)
    from positionbaseddynamics_amd.ensemble import checksum
    sums = ens.gather_checksums([checksum(x)], world)
    replicas_identical = len(set(sums)) == 1

    # PCIe-inclusive rate of the TimeStep::step contract (host ParticleData in -> step -> host ParticleData
    # out every step); reported for information only, never `value`
    t_pcie = t_pcie_pinned = None
    if rank == 0:
        ts.step(model)
        t1 = time.perf_counter()
        for _ in range(3):
            ts.step(model)
        t_pcie = (time.perf_counter() - t1) / 3
        sol.set_option(pbd.Solver.OPT_PIN_HOST, 1)      # page-locked host mirror: transfers at full PCIe rate
        ts.step(model)
        t1 = time.perf_counter()
        for _ in range(3):
            ts.step(model)
        t_pcie_pinned = (time.perf_counter() - t1) / 3
        sol.set_option(pbd.Solver.OPT_PIN_HOST, 0)

    contact_info = None
    if rank == 0 and args.contacts:
        # a floor box under the sheet and a sphere it falls onto; identity collider frames
        eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
        sol.set_colliders([dict(shape="box", params=[50.0, 0.5, 50.0], com=[0, -6.0, 0], R=eye, v1=[0, 0, 0], v2=[0, -6.0, 0], restitution=0.6, friction=0.2),
                           dict(shape="sphere", params=[2.0], com=[5.0, -2.0, 0.0], R=eye, v1=[0, 0, 0], v2=[5.0, -2.0, 0.0], restitution=0.6, friction=0.1)])
        sol.set_collision_ranges([(0, n_particles, 0.6, 0.1)])
        sol.set_contact_params(0.05, 100.0, 5)
        ts.stepResident(model, 3)
        t1 = time.perf_counter()
        ts.stepResident(model, args.steps)
        torch.cuda.synchronize()
        t_c = (time.perf_counter() - t1) / args.steps
        contact_info = {"ms_per_step_with_contact_pass": 1e3 * t_c, "contacts_last_step": sol.num_contacts(),
                        "colliders": 2, "note": "detection + 5 velocity sweeps for every particle against 2 static colliders, once per step"}
        sol.set_colliders([])
        sol.set_collision_ranges([])

    out = {
        "metric": "constraint-projections/s", "value": value, "unit": "projections/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_substep": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_desc,
                   "particles": n_particles, "constraints": n_constraints, "colour_groups": n_groups,
                   "projections_per_substep": projections_per_step, "parallelism": "ensemble x%d (independent instances per GPU, no cross-GPU constraints, no data-path collective)" % world,
                   "state_ok": ok, "replicas_bit_identical": replicas_identical, "state_checksum": "%016x" % sums[0], "device_event_ms_per_substep": stats["total_ms"] / max(args.steps, 1),
                   "algorithmic_GB_per_substep": stats["algorithmic_bytes"] / max(args.steps, 1) / 1e9,
                   "whole_substep_algorithmic_GBs": stats["algorithmic_bytes"] / max(stats["total_ms"], 1e-9) / 1e6,
                   "host_scene_build_s": t_build, "contacts": contact_info, "pcie_inclusive_ms_per_step": None if t_pcie is None else 1e3 * t_pcie,
                   "pcie_inclusive_pinned_ms_per_step": None if t_pcie_pinned is None else 1e3 * t_pcie_pinned, "plan": plan, "persistent": persist, "engine": sol.describe()},
    }

    if rank == 0 and not args.no_roofline:
        # kernel durations measured live with HIP events on the engine's own stream (eager launches, one
        # event in front of every projection launch of the timed configuration)
        sol.set_profiling(True)
        psteps = max(2, min(5, args.steps))
        ts.stepResident(model, psteps)
        sol.set_profiling(False)
        T = pbd.ConstraintType
        pinfo = sol.persistent_info()
        if plan["active"] and pinfo["active"] and pinfo["profiled_launches"]:
            # one launch per substep runs all `iters` sweeps: algorithmic bytes of a launch = iters x bytes of a sweep
            dur_s = 1e-3 * pinfo["profiled_ms"] / pinfo["profiled_launches"]
            bytes_per_launch = pinfo["algorithmic_bytes_per_sweep"] * args.iters
            # with an even number of passes the launch also integrates and updates the velocities (140 B per particle)
            folded = (args.iters * plan["num_segments"]) % 2 == 0
            if folded:
                bytes_per_launch += n_particles * 140
            segs = [sol.segment_info(i) for i in range(plan["num_segments"])]
            streamed = sum(si["stream_bytes"] for si in segs) * args.iters
            achieved = bytes_per_launch / dur_s / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "persistent_kernel (colour-fused LDS tiles, all %d sweeps x %d segments of a substep in one launch%s)" % (
                                   args.iters, plan["num_segments"], ", integration and velocity update included" if folded else ""),
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "streamed_bytes_per_launch": streamed, "avg_launch_us": dur_s * 1e6,
                               "launches_measured": pinfo["profiled_launches"], "grid": pinfo["grid"], "block": pinfo["block"], "lds_bytes": pinfo["lds_bytes"],
                               "segments": [{"segment": i, "colours": [si["colour_begin"], si["colour_end"]], "tiles": si["num_tiles"], "constraints": si["constraints"], "slots": si["slots"],
                                             "algorithmic_bytes_per_pass": si["algorithmic_bytes"], "streamed_bytes_per_pass": si["stream_bytes"]} for i, si in enumerate(segs)],
                               "note": "achieved = SURVEY 8d algorithmic bytes of the launch's distinct constraint projections / event-measured launch time; "
                                       "positions stay in LDS within a pass (and the owned ones between passes), so the bytes actually streamed from HBM (streamed_*) are lower"}
        elif plan["active"]:
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
            # dominant kernel of the per-colour schedule = the constraint type with the largest share of the time
            dom_t, dom_ms = None, 0.0
            for t in range(T.COUNT):
                ms_t, launches_t, _ = sol.type_stats(t)
                if launches_t and ms_t > dom_ms:
                    dom_t, dom_ms = t, ms_t
            if dom_t is not None:
                ms, launches, proj = sol.type_stats(dom_t)
                bytes_per_launch = proj * T.algorithmic_bytes(dom_t) / launches
                dur_s = 1e-3 * ms / launches
                achieved = bytes_per_launch / dur_s / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "project_kernel<%s>" % T.name(dom_t), "achieved": achieved, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                   "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": dur_s * 1e6,
                                   "launches_measured": launches, "per_type": per_type}

    if rank == 0 and world == 1 and "roofline" in out and not args.no_traffic and not args.pmc_child:
        child = ["--workload", args.workload, "--size", str(args.size), "--iters", str(args.iters), "--instances", str(args.instances),
                 "--solid-method", str(args.solid_method)] + (["--bars"] if args.bars else [])
        for flag, val in (("--fuse", args.fuse), ("--tile", args.tile), ("--fuse-block", args.fuse_block), ("--max-seg", args.max_seg),
                          ("--lds-particles", args.lds_particles), ("--xcd-remap", args.xcd_remap), ("--block", args.block),
                          ("--persistent", 2 if persist["active"] else 0)):
            if val is not None:
                child += [flag, str(val)]
        kname = "persistent_kernel" if persist["active"] else "fused_kernel" if plan["active"] else "project_kernel"
        tr = collect_traffic(child, kname)
        if tr is not None:
            r = out["roofline"]
            if plan["active"] and not persist["active"]:
                # the PMC mean runs over the launches of ALL segments: compare with the mean over segments
                rows = r["segments"]
                nl = sum(x["launches"] for x in rows)
                r["traffic_scope"] = "mean over the launches of all %d segments of a sweep" % len(rows)
                r["algorithmic_bytes_per_launch_mean"] = sum(x["algorithmic_bytes_per_launch"] * x["launches"] for x in rows) / nl
                r["streamed_bytes_per_launch_mean"] = sum(x["streamed_bytes_per_launch"] * x["launches"] for x in rows) / nl
            r["traffic"] = tr["bytes_per_launch"]
            r["traffic_detail"] = {k: tr[k] for k in ("fetch_bytes", "write_bytes", "raw")}
        kd = rocprof_kernel_durations(child, kname)
        if kd is not None:
            r = out["roofline"]
            r["rocprofv3_mean_kernel_us"] = kd[0] / 1e3
            r["rocprofv3_dispatches"] = kd[1]
            if plan["active"] and not persist["active"]:
                rows = r["segments"]
                nl = sum(x["launches"] for x in rows)
                r["event_mean_launch_us_all_segments"] = sum(x["avg_us"] * x["launches"] for x in rows) / nl

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size if args.workload != "c3" else 400, args.iters)

    if rank == 0:
        print(json.dumps(out), flush=True)
    ens.close()


if __name__ == "__main__":
    main()
