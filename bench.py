#!/usr/bin/env python3
"""bench.py -- constraint-projections/s and ms/substep on the BASELINE.json headline config:
one 1000x1000 cloth (1 000 000 particles; 2 996 001 XPBD distance + 2 992 005 XPBD isometric
bending constraints; 27 colour groups), 10 solver iterations, 1 substep per step, per GPU.

A "step" is one substep of the hot path (integrate -> 10 x colour-ordered projection sweeps ->
velocity update) over the whole sheet, device-resident (inputs already in HBM when the timed
region starts).

Multi-GPU: a single sheet does not shard; what shards is the ensemble of independent
instances.  `--gpus N` = one process per GPU, no data-path collective; RCCL carries the barrier, the
max-over-ranks time, the counts and per-rank checksums only.  Launched either by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment) or, when WORLD_SIZE is not set, by this script itself:
`python bench.py --gpus N` spawns the N ranks and prints their one line.
    --workload c2 (default)   every rank owns one 1000x1000 sheet                         (weak scaling)
    --workload c4             BASELINE configs[3]: --instances 200x200 sheets per GPU     (weak scaling)
    --workload c4 --scaling strong --total-instances 512
                              the 512 instances sharded contiguously over the ranks       (strong scaling)
value = projections of all ranks / max-over-ranks time.

Output (rank 0).  The LAST stdout line is the compact headline record (< 4 KB: the driver keeps an 8 KB tail):
metric / value / ms_per_step, `config` (workload, counts, per-rank times, `parity_vs_reference` summary), `roofline`,
`cpu_baseline` (N = 1 only).  Before it, one compact line per extra workload ({"extra": ...}: configs[2] variants,
the configs[3] block, the configs[4]-shaped contact scene).  The full record (segment tables, counter raw values, notes,
engine descriptions) goes to `bench_detail.json` next to this script (and to $PBDX_BENCH_DETAIL if set).
`--dry-line [full.json]` prints the compact lines for a stored full record without a GPU (tests/test_bench_line.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0   # the same guide: float4 copy ceiling
CALIB_BYTES = 1 << 28


# ---------------------------------------------------------------------------------------------------
# workloads (scene specifications: positionbaseddynamics_amd/scenes.py)
# ---------------------------------------------------------------------------------------------------
def workload_spec(w, ens):
    """w: dict(workload, size, instances, bars, solid_method, iters, scaling, total_instances).
    Returns (ops, description, pinned particle ids of instance 0)."""
    from positionbaseddynamics_amd import scenes
    if w["workload"] == "c3":
        k = w["instances"] if w["bars"] else 1
        ops = scenes.bar_spec(101, 21, 11, w["solid_method"], instances=k, instanced=True)
        desc = "configs[2]: %s101x21x11 regular tet bar (100000 tets), solid method %d (%s), %d iterations, 1 substep, h=0.005" % (
            ("%d independent bars, each a " % k) if k > 1 else "", w["solid_method"],
            {2: "FEM tets", 3: "XPBD FEM tets", 4: "strain tets", 6: "XPBD distance + volume"}.get(w["solid_method"], "?"), w["iters"])
        return ops, desc, [0]
    if w["workload"] == "c4":
        if w["scaling"] == "strong":
            begin, end = ens.shard(w["total_instances"])
        else:
            begin, end = ens.shard(w["instances"] * ens.world)     # weak scaling: `instances` per GPU
        k = end - begin
        # one prototype + SimulationModel.addInstances: built, coloured and planned once, replicated
        # this rank's instances are the GLOBAL instances begin .. end-1 of the job (their own translations), not 0 .. k-1
        ops = scenes.cloth_spec(w["size"], w["size"], 4, 3, instances=k, instance_offset=(0.0, 0.0, 12.0), instanced=True, first_instance=begin)
        desc = "configs[3]: %d independent %dx%d cloth instances on this GPU (global instances %d..%d; %s; XPBD distance + XPBD isometric bending), %d iterations, 1 substep, h=0.005" % (
            k, w["size"], w["size"], begin, end - 1, ("strong scaling: %d instances over %d GPUs" % (w["total_instances"], ens.world)) if w["scaling"] == "strong" else "weak scaling", w["iters"])
        return ops, desc, [0, w["size"] - 1]
    ops = scenes.cloth_spec(w["size"], w["size"], 4, 3)
    desc = "configs[1]: single %dx%d cloth sheet per GPU (XPBD distance k=1e5 + XPBD isometric bending k=100), %d iterations, 1 substep, h=0.005" % (
        w["size"], w["size"], w["iters"])
    return ops, desc, [0, w["size"] - 1]


def child_flags(w, fused_active, persist_active, opts):
    """Command line that reproduces workload `w` with the same schedule in a profiled child run of this script; the
    schedule the parent measured its way to is FORCED in the child, so that no autotune launch shows up in its trace."""
    child = ["--workload", w["workload"], "--size", str(w["size"]), "--iters", str(w["iters"]), "--instances", str(w["instances"]),
             "--solid-method", str(w["solid_method"])] + (["--bars"] if w["bars"] else [])
    for flag, val in (("--fuse", 1 if fused_active else 0), ("--tile", opts.get("tile")), ("--fuse-block", opts.get("fuse_block")), ("--max-seg", opts.get("max_seg")),
                      ("--lds-particles", opts.get("lds_particles")), ("--xcd-remap", opts.get("xcd_remap")), ("--block", opts.get("block")),
                      ("--wgs-per-cu", opts.get("wgs_per_cu")),
                      ("--persistent", 2 if persist_active else 0)):
        if val is not None:
            child += [flag, str(val)]
    return child


# ---------------------------------------------------------------------------------------------------
# rocprofv3 child passes (counters and traces are never combined; the profiled run is never the timed one)
# ---------------------------------------------------------------------------------------------------
def _rocprof_child(extra_args, child_args, prefix, timeout_s=300, child_steps=None):
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    outdir = tempfile.mkdtemp(prefix=prefix, dir="/tmp")
    cmd = [exe] + extra_args + ["--output-format", "csv", "-d", outdir, "-o", "out", "--",
                                sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args + (
                                    ["--child-steps", str(child_steps)] if child_steps else [])
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
    except Exception:
        shutil.rmtree(outdir, ignore_errors=True)
        return None
    return outdir


def _pmc_pass(counter, child_args):
    """One `rocprofv3 --pmc <counter ...>` pass over a short child run (`counter`: one name, or several that fit one pass: 8 SQ slots, GRBM on its
    own).  One name: returns {kernel_name: [values per dispatch]}; several: {counter: {kernel_name: [...]}}; None if the pass failed."""
    import csv
    import glob
    import shutil
    names = counter.split()
    outdir = _rocprof_child(["--pmc"] + names + ["--kernel-trace"], child_args, "pbdx_pmc_")
    if outdir is None:
        return None
    vals = {n: {} for n in names}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") in vals:
                    vals[row["Counter_Name"]].setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    shutil.rmtree(outdir, ignore_errors=True)
    if len(names) == 1:
        return vals[names[0]] or None
    return vals if any(vals.values()) else None


def rocprof_kernel_durations(child_args, kernel_substring):
    """Cross-check of the event-measured launch durations: a `rocprofv3 --kernel-trace` child pass (no counters) of a
    short run of the same configuration with the schedule FORCED (no autotune launches in the trace): every dispatch
    of the dominant kernel in it is one of the timed kind; the child replays the hipGraph like the timed loop does
    (10 untimed + 30 steps), and the MEDIAN is the figure to compare with ms_per_step (the first dispatches of a process
    run on cold clocks and caches).  Returns dict(median_us, mean_us, min_us, max_us, dispatches)."""
    import csv
    import glob
    import shutil
    outdir = _rocprof_child(["--kernel-trace"], child_args, "pbdx_trace_", child_steps=30)
    if outdir is None:
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
    durs.sort()
    return {"median_us": durs[len(durs) // 2] / 1e3, "mean_us": sum(durs) / len(durs) / 1e3, "min_us": durs[0] / 1e3, "max_us": durs[-1] / 1e3, "dispatches": len(durs)}


def collect_traffic(child_args, kernel_substring):
    """HBM-side bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md (HBM)
    prescribes: FETCH_SIZE and WRITE_SIZE in separate passes; the counters are turned into bytes with
    factors calibrated IN THE SAME PASS on streaming kernels of known size in this engine's own access
    widths (the guide: gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read; other widths
    and WRITE_SIZE must be calibrated).  Infinity-Cache hits are counted as traffic (the guide: these are
    the L2's fabric-side request counters), so this is an UPPER bound of the HBM bytes."""
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
    out["launches"] = res["FETCH_SIZE"]["launches"]
    out["raw"] = res
    return out


# ---------------------------------------------------------------------------------------------------
# one workload on this rank's GPU
# ---------------------------------------------------------------------------------------------------
def run_workload(w, opts, ens, steps, warmup, with_roofline, with_traffic, with_pcie=False, with_contacts=False):
    """Builds the scene, steps it device-resident (untimed warm-up, then EXACTLY `steps` steps bracketed by a barrier +
    device synchronisation on both sides), returns the measurements of this rank plus -- rank 0 -- the roofline record."""
    import torch
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd import scenes
    from positionbaseddynamics_amd.ensemble import checksum
    S = pbd.Solver
    ops, desc, pins = workload_spec(w, ens)
    t0 = time.perf_counter()
    model = scenes.build_model(ops)
    model.initConstraintGroups()
    t_build = time.perf_counter() - t0
    n_particles = model.getParticles().size()
    n_constraints = model.numConstraints()
    n_groups = len(model.getConstraintGroups())
    iters = w["iters"]

    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController(device=ens.hip_device)
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
    sol = ts.solver()
    for key, opt in (("xcd_remap", S.OPT_XCD_REMAP), ("block", S.OPT_BLOCK_SIZE), ("fuse", S.OPT_FUSE), ("tile", S.OPT_TILE_PARTICLES),
                     ("fuse_block", S.OPT_FUSE_BLOCK), ("max_seg", S.OPT_MAX_SEGMENT_COLOURS), ("lds_particles", S.OPT_LDS_PARTICLES),
                     ("wgs_per_cu", S.OPT_PERSISTENT_WGS_PER_CU), ("persistent", S.OPT_PERSISTENT)):
        if opts.get(key) is not None:
            sol.set_option(opt, opts[key])
    if opts.get("no_graph"):
        sol.set_option(S.OPT_USE_GRAPH, 0)

    def barrier():
        torch.cuda.synchronize()
        ens.barrier()

    # warm-up (untimed): uploads the device image, plans, measures the schedule candidates, instantiates the hipGraph
    ts.stepResident(model, max(warmup, 1))
    # the timed region: EXACTLY `steps` steps, nothing else on the stream (the per-substep event records used to sit inside it)
    barrier()
    t0 = time.perf_counter()
    ts.stepResident(model, steps)          # synchronises its own stream before returning
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    stats = sol.stats()
    # "hipEvents around the device-resident substep loop ..., >= 50 substeps, median": an UNTIMED replay of the same `steps` steps right
    # after the timed region with one event after every substep on the engine's stream (the graph replay is unchanged): the median DEVICE time per
    # substep, reported beside the host-clock mean of the timed region that `value` is computed from
    sol.set_option(S.OPT_SUBSTEP_EVENTS, max(int(steps), 1))      # (n > 1: the events are created here)
    ts.stepResident(model, steps)
    torch.cuda.synchronize()
    sub_ms = sorted(sol.substep_times())
    sol.set_option(S.OPT_SUBSTEP_EVENTS, 0)
    substep_device = None
    if sub_ms:
        substep_device = {"n": len(sub_ms), "median_ms": sub_ms[len(sub_ms) // 2], "mean_ms": sum(sub_ms) / len(sub_ms), "min_ms": sub_ms[0],
                          "p90_ms": sub_ms[min(len(sub_ms) - 1, (9 * len(sub_ms)) // 10)], "max_ms": sub_ms[-1]}
    plan = sol.plan_info()
    persist = sol.persistent_info()
    barrier()

    # sanity of the timed state: finite, pinned corners unmoved; checksum for the cross-rank comparison
    ts.syncToHost(model)
    x = model.getParticles().positions()
    x0 = model.getParticles().array(1)
    ok = bool(np.all(np.isfinite(x)) and all(np.array_equal(x[p], x0[p]) for p in pins))
    res = {"desc": desc, "t_local": t_local, "n_particles": n_particles, "n_constraints": n_constraints, "n_groups": n_groups,
           "t_build": t_build, "stats": stats, "plan": plan, "persistent": persist, "state_ok": ok, "checksum": checksum(x),
           "engine": sol.describe(), "steps_done": max(warmup, 1) + 2 * steps, "substep_device": substep_device}

    if with_pcie and ens.rank == 0:
        # PCIe-inclusive rate of the TimeStep::step contract (host ParticleData in -> step -> host ParticleData out every
        # step); reported for information only, never `value`
        ts.step(model)
        t1 = time.perf_counter()
        for _ in range(3):
            ts.step(model)
        res["pcie_ms"] = 1e3 * (time.perf_counter() - t1) / 3
        sol.set_option(S.OPT_PIN_HOST, 1)      # page-locked host mirror: transfers at full PCIe rate
        ts.step(model)
        t1 = time.perf_counter()
        for _ in range(3):
            ts.step(model)
        res["pcie_pinned_ms"] = 1e3 * (time.perf_counter() - t1) / 3
        sol.set_option(S.OPT_PIN_HOST, 0)

    if with_contacts and ens.rank == 0:
        # a floor box under the sheet and a sphere it falls onto; identity collider frames
        eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
        sol.set_colliders([dict(shape="box", params=[50.0, 0.5, 50.0], com=[0, -6.0, 0], R=eye, v1=[0, 0, 0], v2=[0, -6.0, 0], restitution=0.6, friction=0.2),
                           dict(shape="sphere", params=[2.0], com=[5.0, -2.0, 0.0], R=eye, v1=[0, 0, 0], v2=[5.0, -2.0, 0.0], restitution=0.6, friction=0.1)])
        sol.set_collision_ranges([(0, n_particles, 0.6, 0.1)])
        sol.set_contact_params(0.05, 100.0, 5)
        ts.stepResident(model, 3)
        t1 = time.perf_counter()
        ts.stepResident(model, steps)
        torch.cuda.synchronize()
        t_c = (time.perf_counter() - t1) / steps
        res["contacts"] = {"ms_per_step_with_contact_pass": 1e3 * t_c, "contacts_last_step": sol.num_contacts(),
                           "colliders": 2, "note": "detection + 5 velocity sweeps for every particle against 2 static colliders, once per step"}
        sol.set_colliders([])
        sol.set_collision_ranges([])

    if with_roofline and ens.rank == 0:
        res["roofline"] = roofline_record(pbd, sol, ts, model, w, plan, steps, n_particles, stats, substep_device)
        if res["roofline"] is not None:
            # until a counter pass replaces them, achieved / frac are SURVEY 8d's algorithmic-byte figures (can exceed 1: LDS-resident tiles)
            res["roofline"]["frac_kind"] = "contract"
        if with_traffic and res["roofline"] is not None and ens.world == 1:
            add_profiled_passes(res["roofline"], w, opts, plan, persist)
    return res


def roofline_record(pbd, sol, ts, model, w, plan, steps, n_particles, timed_stats=None, substep_device=None):
    """Dominant kernel of the workload, bytes as SURVEY 8d defines them, duration measured live with HIP events on the
    engine's own stream.  Where ONE launch is the whole substep (persistent schedule with integration and velocity update
    folded in) the duration is taken in the SAME mode as the timed loop: the device-event time of the timed region (graph
    replay, `steps` launches back to back) / `steps` -- an upper bound of the kernel's own duration (it includes the gaps
    between graph launches), so that kernel time <= ms_per_step by construction; the eager event-pair figure (one pair
    around a single launch on an otherwise idle GPU: slower, cold clocks and caches) is kept as `eager_launch_us`.
    Other schedules: one event pair around every projection launch (eager)."""
    iters = w["iters"]
    sol.set_profiling(True)
    psteps = max(2, min(5, steps))
    ts.stepResident(model, psteps)
    sol.set_profiling(False)
    T = pbd.ConstraintType
    pinfo = sol.persistent_info()
    if plan["active"] and pinfo["active"] and pinfo["profiled_launches"]:
        # one launch per substep runs all `iters` sweeps: algorithmic bytes of a launch = iters x bytes of a sweep
        eager_s = 1e-3 * pinfo["profiled_ms"] / pinfo["profiled_launches"]
        dur_s, mode = eager_s, "eager launch, HIP event pair around it"
        bytes_per_launch = pinfo["algorithmic_bytes_per_sweep"] * iters
        folded = bool(pinfo["last_folded"])
        if folded and timed_stats and timed_stats["total_ms"] > 0 and timed_stats["kernel_launches"] == steps:
            dur_s = 1e-3 * timed_stats["total_ms"] / steps
            mode = "timed region: HIP events around the %d graph-replayed launches of the timed loop / %d" % (steps, steps)
            if substep_device and substep_device["n"] == steps:
                dur_s = 1e-3 * substep_device["median_ms"]
                mode = "timed region: MEDIAN of the %d per-substep HIP-event intervals of the timed loop (one launch per substep, graph replay)" % steps
        if folded:      # the launch also integrates and updates the velocities (140 B per particle)
            bytes_per_launch += n_particles * 140
        segs = [sol.segment_info(i) for i in range(plan["num_segments"])]
        streamed = sum(si["stream_bytes"] for si in segs) * iters
        compulsory = plan["compulsory_stream_bytes_per_sweep"] * iters + n_particles * 140
        achieved = bytes_per_launch / dur_s / 1e9
        return {"bound": "hbm", "kernel": "persistent_kernel: %d sweeps x %d segments%s in ONE launch (colour-fused LDS tiles)" % (
                    iters, plan["num_segments"], " + integrate + velocity update" if folded else ""),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "streamed_bytes_per_launch": streamed,
                "compulsory_bytes_per_launch": compulsory, "avg_launch_us": dur_s * 1e6, "timing_mode": mode,
                "mean_launch_us_timed_region": (1e3 * timed_stats["total_ms"] / steps) if (timed_stats and steps) else None,
                "eager_launch_us": eager_s * 1e6,
                "launches_measured": steps if dur_s != eager_s else pinfo["profiled_launches"], "grid": pinfo["grid"], "block": pinfo["block"], "lds_bytes": pinfo["lds_bytes"], "folded": folded,
                "segments": [{"segment": i, "colours": [si["colour_begin"], si["colour_end"]], "tiles": si["num_tiles"], "constraints": si["constraints"], "slots": si["slots"],
                              "algorithmic_bytes_per_pass": si["algorithmic_bytes"], "streamed_bytes_per_pass": si["stream_bytes"]} for i, si in enumerate(segs)],
                "note": "frac = SURVEY 8d ALGORITHMIC bytes (every endpoint position read and written once per projection, 32-bit indices, no cache credit) / "
                        "event-measured launch time / 8 TB/s: it can exceed 1 because the kernel keeps positions in LDS and streams 16-bit indices. "
                        "frac_traffic = counter-measured fabric bytes (upper bound of HBM bytes: Infinity-Cache hits are counted) / the same time / 8 TB/s is the "
                        "physically bounded figure; compulsory_bytes = every distinct constraint record once per sweep + one particle-state pass; "
                        "traffic_over_compulsory is the redundancy left to remove"}
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
        if not rows:
            return None
        dom = max(rows, key=lambda r: r["avg_us"] * r["launches"])
        return {"bound": "hbm", "kernel": "fused_kernel (colour-fused LDS tiles), segment %d = colours [%d,%d)" % (dom["segment"], dom["colours"][0], dom["colours"][1]),
                "achieved": dom["algorithmic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["algorithmic_GBs"] / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "streamed_bytes_per_launch": dom["streamed_bytes_per_launch"], "avg_launch_us": dom["avg_us"],
                "launches_measured": dom["launches"], "segments": rows,
                "note": "achieved = SURVEY 8d algorithmic bytes of the segment's distinct constraints / event-measured launch time; "
                        "positions stay in LDS for the whole segment, so the bytes actually streamed from HBM (streamed_*) are lower"}
    per_type = {}
    dom_t, dom_ms = None, 0.0
    for t in range(T.COUNT):
        ms, launches, proj = sol.type_stats(t)
        if launches:
            per_type[T.name(t)] = {"launches": launches, "avg_us": 1e3 * ms / launches, "projections": proj,
                                   "algorithmic_GBs": proj * T.algorithmic_bytes(t) / ms / 1e6}
            if ms > dom_ms:
                dom_t, dom_ms = t, ms
    if dom_t is None:
        return None
    ms, launches, proj = sol.type_stats(dom_t)
    bytes_per_launch = proj * T.algorithmic_bytes(dom_t) / launches
    dur_s = 1e-3 * ms / launches
    achieved = bytes_per_launch / dur_s / 1e9
    return {"bound": "hbm", "kernel": "project_kernel<%s>" % T.name(dom_t), "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": dur_s * 1e6,
            "launches_measured": launches, "per_type": per_type}


def add_profiled_passes(r, w, opts, plan, persist):
    """roofline.traffic (PMC child passes) and the rocprofv3 kernel-trace cross-check of the event-measured duration."""
    child = child_flags(w, plan["active"], persist["active"], opts)
    kname = "persistent_kernel" if persist["active"] else "fused_kernel" if plan["active"] else "project_kernel"
    tr = collect_traffic(child, kname)
    if tr is not None:
        if plan["active"] and not persist["active"]:
            # the PMC mean runs over the launches of ALL segments: compare with the mean over segments
            rows = r["segments"]
            nl = sum(x["launches"] for x in rows)
            r["traffic_scope"] = "mean over the launches of all %d segments of a sweep" % len(rows)
            r["algorithmic_bytes_per_launch_mean"] = sum(x["algorithmic_bytes_per_launch"] * x["launches"] for x in rows) / nl
            r["streamed_bytes_per_launch_mean"] = sum(x["streamed_bytes_per_launch"] * x["launches"] for x in rows) / nl
        r["traffic"] = tr["bytes_per_launch"]
        r["traffic_detail"] = {k: tr[k] for k in ("fetch_bytes", "write_bytes", "launches", "raw")}
    kd = rocprof_kernel_durations(child, kname)
    if kd is not None:
        r["rocprofv3_median_kernel_us"] = kd["median_us"]
        r["rocprofv3_mean_kernel_us"] = kd["mean_us"]
        r["rocprofv3_min_kernel_us"] = kd["min_us"]
        r["rocprofv3_max_kernel_us"] = kd["max_us"]
        r["rocprofv3_dispatches"] = kd["dispatches"]
        if plan["active"] and not persist["active"]:
            rows = r["segments"]
            nl = sum(x["launches"] for x in rows)
            r["event_mean_launch_us_all_segments"] = sum(x["avg_us"] * x["launches"] for x in rows) / nl
    dur_us = r["avg_launch_us"] if persist["active"] else r.get("event_mean_launch_us_all_segments") if plan["active"] else None
    if persist["active"] and dur_us:
        # VALU issue: SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD issued a vector-ALU instruction, summed over the SIMDs) against the
        # SIMD cycles of the launch (GRBM_GUI_ACTIVE: busy cycles summed over the 8 XCDs) -- clock-free
        # (until round 4 this was SQ_ACTIVE_INST_VALU x 4 cycles -- but a SIMD issues a wave64 fp32 instruction every ~2 cycles when it
        # has four waves to choose from; the interval is MEASURED here, in this run, at the kernel's own occupancy: pbdx_debug_valu_issue)
        pm = _pmc_pass("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE", child) or {}

        def mean_of(name):
            xs = [x for k, v in (pm.get(name) or {}).items() if kname in k for x in v]
            return sum(xs) / len(xs) if xs else None
        iv, va, g = mean_of("SQ_INSTS_VALU"), mean_of("SQ_ACTIVE_INST_VALU"), mean_of("GRBM_GUI_ACTIVE")
        if iv and g:
            import positionbaseddynamics_amd as pbd
            simds = 4 * 256
            r["valu_instructions_per_launch"] = iv
            r["valu_active_quad_cycles_per_launch"] = va
            r["gpu_cycles_per_launch"] = g / 8.0
            cpi = pbd.valu_issue_interval(0, int(persist.get("block") or 1024))
            r["valu_cycles_per_instruction_measured"] = cpi
            r["frac_valu"] = iv * cpi / (simds * r["gpu_cycles_per_launch"])
            r["frac_valu_definition"] = ("SQ_INSTS_VALU x the measured issue interval of a SIMD at this kernel's occupancy (pbdx_debug_valu_issue: v_mul_f32 / v_add_f32, "
                                         "%d threads per workgroup, one workgroup per CU) / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8)" % int(persist.get("block") or 1024))
            la, lc, li = mean_of("SQ_LDS_IDX_ACTIVE"), mean_of("SQ_LDS_BANK_CONFLICT"), mean_of("SQ_INSTS_LDS")
            if la:
                r["lds_active_cycles_per_launch"] = la
                r["lds_bank_conflict_cycles_per_launch"] = lc
                r["lds_instructions_per_launch"] = li
                r["frac_lds"] = la / (256 * r["gpu_cycles_per_launch"])      # LDS-array cycles of a CU / the launch's cycles
    if r.get("traffic") and dur_us:
        dur_s = dur_us * 1e-6
        r["traffic_GBs"] = r["traffic"] / dur_s / 1e9
        r["frac_traffic"] = r["traffic_GBs"] / HBM_PEAK_GBS
        try:
            import positionbaseddynamics_amd as pbd
            r["copy_ceiling_gbs"] = pbd.copy_bandwidth(0)      # the practical roof, measured in this run (float4 copy, 1 GiB per direction)
        except Exception:
            r["copy_ceiling_gbs"] = None
        r["copy_ceiling_gbs_guide"] = HBM_COPY_GBS
        r["frac_traffic_of_copy_ceiling"] = r["traffic_GBs"] / (r["copy_ceiling_gbs"] or HBM_COPY_GBS)
        # the distance to the machine as ONE number: counter traffic / time against the guide's measured copy bandwidth, and the time the
        # compulsory bytes would take at the copy bandwidth measured in THIS run -- what a launch that did nothing but move its bytes would need
        r["frac_of_achievable"] = r["traffic_GBs"] / HBM_COPY_GBS
        if r.get("compulsory_bytes_per_launch"):
            r["floor_ms"] = 1e3 * r["compulsory_bytes_per_launch"] / ((r["copy_ceiling_gbs"] or HBM_COPY_GBS) * 1e9)
            r["time_over_floor"] = (dur_s * 1e3) / r["floor_ms"]
        r["traffic_over_algorithmic"] = r["traffic"] / r.get("algorithmic_bytes_per_launch_mean", r["algorithmic_bytes_per_launch"])
        if r.get("compulsory_bytes_per_launch"):
            r["traffic_over_compulsory"] = r["traffic"] / r["compulsory_bytes_per_launch"]
            r["frac_compulsory"] = r["compulsory_bytes_per_launch"] / dur_s / 1e9 / HBM_PEAK_GBS
        # achieved / frac stay what the contract defines on EVERY line (SURVEY 8d algorithmic bytes / device time / 8 TB/s; ADVICE r4: they used to switch
        # meaning with the counter pass); the physically bounded figures are keys of their own: achieved_traffic / frac_traffic (counter-measured fabric
        # bytes), frac_traffic_of_copy_ceiling (against the copy bandwidth measured in this run), frac_valu, frac_lds
        r["achieved_contract"], r["frac_contract"] = r["achieved"], r["frac"]
        r["achieved_traffic"] = r["traffic_GBs"]
        r["note"] = ("achieved / frac = SURVEY 8d ALGORITHMIC bytes (every endpoint position read and written once per projection, 32-bit indices, no cache credit) / "
                     "device time / 8 TB/s: it exceeds 1 because the kernel keeps positions in LDS and streams 16-bit indices (the work is proven bit-identical). "
                     "achieved_traffic / frac_traffic = counter-measured fabric bytes per launch (upper bound of HBM bytes: Infinity-Cache hits are counted) / the same "
                     "time / 8 TB/s -- the physically bounded figure; frac_valu / frac_lds = share of the launch's cycles a SIMD needs to issue its vector-ALU instructions "
                     "at the measured rate / the CU's LDS array is busy. compulsory_bytes = every distinct constraint record once per sweep + one particle-state pass")
        r["frac_definition"] = "SURVEY 8d algorithmic bytes per launch / device time per launch / 8 TB/s; frac_traffic = counter-measured HBM-side bytes (FETCH_SIZE + WRITE_SIZE, calibrated in-pass) / same time / 8 TB/s"
    if r.get("frac_valu") is not None:
        fr = {"hbm traffic": r.get("frac_traffic") or 0.0, "valu issue": r["frac_valu"], "lds array": r.get("frac_lds") or 0.0}
        top = max(fr, key=fr.get)
        r["binding"] = ("none of the three units is saturated (%s): what is left of the launch is synchronisation -- workgroup barriers between dependent colour steps, "
                        "record-fetch issue, tile hand-offs at the pass boundaries" % ", ".join("%s %.2f" % kv for kv in sorted(fr.items(), key=lambda kv: -kv[1]))
                        ) if fr[top] < 0.8 else "%s (%.2f of its peak)" % (top, fr[top])


# ---------------------------------------------------------------------------------------------------
# CPU side: the reference itself on this box's host cores (timing) and as the checker of the timed engine (parity)
# ---------------------------------------------------------------------------------------------------
_NPROC_LEG = r"""
import json, sys, time
sys.path.insert(0, %(root)r)
from oracle import refdrv
from oracle.scene_ref import apply_ref
from positionbaseddynamics_amd import scenes
o = refdrv.Ref(%(variant)r)
apply_ref(o, scenes.cloth_spec(%(size)d, %(size)d, 4, 3))
o.set_time_step_size(0.005)
o.set_params(1, %(iters)d, 0)
o.set_num_threads(64)
o.step(1)
o.set_num_threads(%(ncpu)d)
t = o.time_steps(1)
print(json.dumps({"threads": %(ncpu)d, "ms_per_substep": 1e3 * t, "projections_per_s": o.num_constraints() * %(iters)d / t, "timed_steps": 1}))
"""


def cpu_nproc_leg(size, iters, ncpu, variant, limit_s=150):
    """One step of the reference at OMP_NUM_THREADS = nproc in a child process (see cpu_baseline); a dict, with `timed_out_after_s` if it did not finish."""
    import subprocess
    try:
        cp = subprocess.run([sys.executable, "-c", _NPROC_LEG % {"root": os.path.dirname(os.path.abspath(__file__)), "variant": variant, "size": size, "iters": iters, "ncpu": ncpu}],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s)
        if cp.returncode == 0 and cp.stdout.strip():
            return json.loads(cp.stdout.strip().splitlines()[-1])
        return {"threads": ncpu, "error": (cp.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"threads": ncpu, "timed_out_after_s": limit_s}


def cpu_baseline(size, iters, opts, hip_device, parity_steps=3, timed_steps=3):
    """(1) TIMING.  The reference's own TimeStepController::step (oracle/_ref, release-like build: -O3 -fopenmp and the
    widest -march level this host executes, oracle/refdrv.py:best_timing_variant) on the SAME scene as the GPU workload
    (size x size cloth, XPBD distance + XPBD isometric bending, same iteration count): `timed_steps` steps at 1 OpenMP
    thread and at a few multi-thread settings after a warm-up step (which includes the one-off colouring).  The reference
    forks/joins one parallel region per colour group, so large thread counts can be SLOWER; the best setting is `value`.
    (2) PARITY.  The contraction-free float build of the reference (oracle/_ref f32: the build the engine is bit-identical
    to; the release-like build contracts a*b+c into fused multiply-adds) steps the same scene `parity_steps` steps on the host;
    the GPU engine, with the options of the timed run, steps a fresh copy of the scene the same number of steps; all
    positions and velocities are compared bit for bit.  Returns (cpu_baseline record, parity record)."""
    try:
        from oracle import refdrv
        from oracle.scene_ref import apply_ref
        from positionbaseddynamics_amd import scenes
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "projections/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}, None
    ncpu = os.cpu_count() or 1
    ops = scenes.cloth_spec(size, size, 4, 3)
    variant = refdrv.best_timing_variant()
    if variant is None:
        from oracle import port
        o = port.Port("f32")
        kind, thread_settings, flags = "port", [1], "plain-C port, gcc -O2"
    else:
        o = refdrv.Ref(variant)
        kind = "reference"
        # SURVEY 8d asks for OMP_NUM_THREADS = 1 and = nproc: at nproc = 256 (the MI355X box's host) the reference needs 57.8 s per substep -- it forks and joins
        # one parallel region per colour group (profiles/r04_bench_detail.json of the run that tried it: 4 minutes of bench time) -- so the sweep stops at 64 and
        # the best setting is reported
        thread_settings = sorted(set([1, min(16, ncpu), min(64, ncpu)]))
        flags = {"v4": "-O3 -march=x86-64-v4 -fopenmp, float", "fast": "-O3 -march=x86-64-v3 -fopenmp, float"}[variant]
    t_setup = time.perf_counter()
    apply_ref(o, ops)
    o.set_time_step_size(0.005)
    o.set_params(1, iters, 0)
    nc = o.num_constraints()
    o.set_num_threads(max(thread_settings))
    o.step(1)   # warm-up: includes the one-off colouring
    t_setup = time.perf_counter() - t_setup
    results = {}
    for th in reversed(thread_settings):      # multi-thread settings first: the 1-thread leg is the long one
        o.set_num_threads(th)
        t = 0.0
        n = timed_steps if th > 1 else max(1, timed_steps - 1)      # (the 1-thread leg is 5-7 s per step: one step fewer keeps the default run near 90 s)
        for _ in range(n):
            t += o.time_steps(1)
        results[th] = (nc * iters * n / t, 1e3 * t / n, n)
    best = max(results, key=lambda k: results[k][0])
    rec = {"value": results[best][0], "unit": "projections/s", "cores": best, "kind": kind,
           "ms_per_substep": results[best][1], "variant": variant, "host_logical_cpus": ncpu,
           "single_thread": {"projections_per_s": results[1][0], "ms_per_substep": results[1][1]} if 1 in results else None,
           "by_threads": {str(k): {"projections_per_s": v[0], "ms_per_substep": v[1], "timed_steps": v[2]} for k, v in sorted(results.items())},
           "sample": "the GPU workload itself: %dx%d cloth (%d constraints), %d iterations x 1 substep; %d timed steps per thread setting (one fewer at 1 thread) after one warm-up step; "
                     "reference build '%s' (%s); OMP threads tried %s, best = %d (host reports %d logical CPUs); scene build + colouring + warm-up %.1fs" % (
                         size, size, nc, iters, timed_steps, variant, flags, sorted(results), best, ncpu, t_setup)}
    o.reset_all()
    # BASELINE.md 2 / SURVEY 8d also ask for OMP_NUM_THREADS = nproc.  On the MI355X box's 256-CPU host that leg takes about a minute per substep (one
    # fork / join per colour group and iteration), so it is ONE step in a child process with a time limit; reported as a field, never `value`
    rec["nproc_threads"] = None
    # (opt-in since round 6, --cpu-nproc: informative once -- 60 s of every driver run for one step -- and on record in profiles/r05_bench_detail.json)
    if variant is not None and ncpu > max(thread_settings) and opts.get("cpu_nproc"):
        rec["nproc_threads"] = cpu_nproc_leg(size, iters, ncpu, variant)

    parity = None
    if refdrv.available("f32") and parity_steps > 0:
        import positionbaseddynamics_amd as pbd
        S = pbd.Solver
        ref = refdrv.Ref("f32")
        apply_ref(ref, ops)
        ref.set_time_step_size(0.005)
        ref.set_gravity(scenes.GRAVITY)
        ref.set_params(1, iters, 0)
        ref.set_num_threads(min(32, ncpu))
        t0 = time.perf_counter()
        ref.step(parity_steps)
        t_ref = time.perf_counter() - t0
        xr, vr = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
        ref.reset_all()
        model = scenes.build_model(ops)
        pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
        ts = pbd.TimeStepController(device=hip_device)
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
        sol = ts.solver()
        for key, opt in (("xcd_remap", S.OPT_XCD_REMAP), ("fuse", S.OPT_FUSE), ("tile", S.OPT_TILE_PARTICLES), ("fuse_block", S.OPT_FUSE_BLOCK),
                         ("max_seg", S.OPT_MAX_SEGMENT_COLOURS), ("lds_particles", S.OPT_LDS_PARTICLES), ("persistent", S.OPT_PERSISTENT)):
            if opts.get(key) is not None:
                sol.set_option(opt, opts[key])
        ts.stepResident(model, parity_steps)
        ts.syncToHost(model)
        xg, vg = model.getParticles().positions(), model.getParticles().array(2)
        pi = sol.persistent_info()
        same = bool(np.array_equal(xg.view(np.uint32), xr.view(np.uint32)) and np.array_equal(vg.view(np.uint32), vr.view(np.uint32)))
        parity = {"scene": "%dx%d cloth, %d iterations x 1 substep (the timed workload)" % (size, size, iters), "steps": parity_steps,
                  "reference": "oracle/_ref f32 (unmodified reference sources, -O2 -ffp-contract=off), %d OpenMP threads, %.1f s" % (min(32, ncpu), t_ref),
                  "bit_identical": same, "max_abs": float(np.max(np.abs(xg.astype(np.float64) - xr.astype(np.float64)))),
                  "max_abs_velocity": float(np.max(np.abs(vg.astype(np.float64) - vr.astype(np.float64)))),
                  "compared_values": int(xg.size + vg.size),
                  "engine_schedule": {"fused": sol.plan_info()["active"], "persistent": pi["active"], "folded": pi["last_folded"], "refusals": pi["refusals"], "timeouts": pi["timeouts"]}}
    return rec, parity


def extra_parity(ew, ens, hip_device, steps=2):
    """Parity leg of an extra workload (`state_ok` only says finite + pins): the scene stepped `steps` steps from rest by the reference's
    contraction-free float build (oracle/_ref f32, up to 32 threads) and by a fresh engine with default options; positions and velocities bit for bit."""
    try:
        from oracle import refdrv
        from oracle.scene_ref import apply_ref
        import positionbaseddynamics_amd as pbd
        from positionbaseddynamics_amd import scenes
        if not refdrv.available("f32"):
            return None
        ops, _, _ = workload_spec(ew, ens)
        ncpu = os.cpu_count() or 1
        t0 = time.perf_counter()
        ref = refdrv.Ref("f32")
        apply_ref(ref, ops)
        ref.set_time_step_size(0.005)
        ref.set_gravity(scenes.GRAVITY)
        ref.set_params(1, ew["iters"], 0)
        ref.set_num_threads(min(32, ncpu))
        ref.step(steps)
        xr, vr = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
        ref.reset_all()
        t_ref = time.perf_counter() - t0
        model = scenes.build_model(ops)
        pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
        ts = pbd.TimeStepController(device=hip_device)
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, ew["iters"])
        ts.stepResident(model, steps)
        ts.syncToHost(model)
        xg, vg = model.getParticles().positions(), model.getParticles().array(2)
        return {"bit_identical": bool(np.array_equal(xg.view(np.uint32), xr.view(np.uint32)) and np.array_equal(vg.view(np.uint32), vr.view(np.uint32))),
                "steps": steps, "from": "rest", "compared_values": int(xg.size + vg.size), "reference_seconds": t_ref,
                "max_abs": float(np.max(np.abs(xg.astype(np.float64) - xr.astype(np.float64))))}
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}


def run_c5(ens, sub_steps, with_reference=True):
    """BASELINE configs[4] in the shape that can be pinned (positionbaseddynamics_amd/scenes.py: armadillo_collision_scene): three armadillo_4k FEM
    solids + static floor, floor contacts and deformable-deformable contacts, maxIterations 1, maxIterationsV 5, h = 0.01.  Timed: 260 steps
    resident in one call.  Parity: a second run, step by step, against the reference's committed results after 120 and 260 steps (positions,
    velocities, contact lists: bit for bit).  CPU beside it: the reference itself on the same scene (oracle/_ref, 1 thread), 60 steps."""
    import torch
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd import scenes
    ops, g = scenes.armadillo_collision_scene()

    def fresh():
        model = scenes.build_model(ops)
        pbd.TimeManager.getCurrent().setTimeStepSize(float(g["time_step"]))
        ts = pbd.TimeStepController(device=ens.hip_device)
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, sub_steps)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, int(g["iterations"]))
        ts.syncFromHost(model)
        sol = ts.solver()
        cols = scenes.install_armadillo_colliders(sol, g)
        return model, ts, sol, cols

    model, ts, sol, cols = fresh()
    ts.stepResident(model, 2)                   # plan, schedule measurement
    model, ts, sol, cols = fresh()
    ts.stepResident(model, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts.stepResident(model, 259)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 259
    engine = sol.describe()
    n_particles, n_constraints = model.getParticles().size(), model.numConstraints()
    # parity pass
    model, ts, sol, cols = fresh()
    same, compared, done, floor, tet_max = True, 0, 0, 0, 0
    for steps in g["steps"]:
        for _ in range(int(steps) - done):
            ts.stepResident(model, 1)
            floor += sol.num_contacts()
            tet_max = max(tet_max, sol.num_tet_contacts())
        done = int(steps)
        ts.syncToHost(model)
        got, want = sol.tet_contacts(), g["contacts_sub%d_%d" % (sub_steps, steps)]
        x, v = model.getParticles().positions(), model.getParticles().velocities()
        same = same and len(got) == len(want) and (not len(want) or np.array_equal(got[:, :26].view(np.uint32), want.view(np.uint32)))
        same = same and np.array_equal(x.view(np.uint32), g["x_sub%d_%d" % (sub_steps, steps)].view(np.uint32)) and np.array_equal(v.view(np.uint32), g["v_sub%d_%d" % (sub_steps, steps)].view(np.uint32))
        compared += x.size + v.size + got[:, :26].size
    totals = g["contact_totals_sub%d" % sub_steps]
    out = {"tag": "c5_armadillo_%dsub" % sub_steps,
           "workload": "configs[4] (pinnable shape): 3 x armadillo_4k FEM tet solids (analytic box SDF stand-in, friction 0) + static floor, floor AND deformable-deformable contacts, "
                       "%d substeps, maxIterations 1, maxIterationsV 5, h=0.01" % sub_steps,
           "particles": n_particles, "constraints": n_constraints, "steps": 260, "sub_steps": sub_steps, "iterations": 1, "ms_per_step": ms, "ms_per_substep": ms / sub_steps,
           "state_ok": bool(np.all(np.isfinite(model.getParticles().positions()))), "bit_identical": bool(same and floor == int(totals[1])), "compared_values": int(compared), "parity_steps": 260,
           "contacts_deformable_total": int(totals[0]), "contacts_floor_total": int(floor), "contacts_max": int(tet_max), "engine": engine}
    if with_reference:
        try:
            from oracle import refdrv
            if refdrv.available("f32"):
                ref = refdrv.Ref("f32")
                ref.reset_all(); ref.set_num_threads(1); ref.set_time_step_size(0.01); ref.set_gravity(scenes.GRAVITY)
                for q in range(3):
                    _, _, offset, nv, nt, _ = (int(v) for v in g["c%d_meta" % q])
                    ref.add_tet_model(g["x0"][offset:offset + nv].astype(np.float64), g["c%d_tets" % q].reshape(-1, 4))
                    ref.set_tet_model_initial_transform(q, g["c%d_initial_x" % q].astype(np.float64), g["c%d_initial_R" % q].astype(np.float64).reshape(3, 3))
                for q in range(3):
                    ref.add_solid_constraints(q, 2, 1.0, 0.2, 1.0, False, False)
                ref.set_collision_tolerance(0.0)
                ref.add_static_collider("box", (0, 0, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.0)
                for q in range(3):
                    ref.add_tet_collision_shape(q, 0, [float(v) for v in g["box"]], True, False, float(g["c%d_restitution" % q]), 0.0)
                ref.attach_collision_detection()
                ref.set_params(sub_steps, 1, 0); ref.set_max_iterations_v(5)
                ref.step(60)
                t1 = time.perf_counter()
                ref.step(60)
                out["reference_ms_per_step"] = 1e3 * (time.perf_counter() - t1) / 60
                # the live reference also reproduces the fixture (steps 120)
                out["reference_reproduces_fixture"] = bool(np.array_equal(ref.positions().astype(np.float32).view(np.uint32), g["x_sub%d_120" % sub_steps].view(np.uint32)))
                ref.reset_all()
        except Exception as e:  # pragma: no cover
            out["reference_error"] = repr(e)
    del cols
    return out


def shard_vs_reference(w, ens, opts, steps):
    """--check-shards (checker leg, small sizes): THIS rank's instances stepped by the reference itself (oracle/_ref f32) and by a
    fresh engine, compared bit for bit.  Ranks hold different instances, so this -- not a comparison of the ranks' checksums with each
    other -- is what says that a shard simulated the instances it was given."""
    from oracle import refdrv
    from oracle.scene_ref import apply_ref
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd import scenes
    ops, _, _ = workload_spec(w, ens)
    ref = refdrv.Ref("f32")
    apply_ref(ref, ops)
    ref.set_time_step_size(0.005)
    ref.set_gravity(scenes.GRAVITY)
    ref.set_params(1, w["iters"], 0)
    ref.set_num_threads(1)
    ref.step(steps)
    xr = ref.positions().astype(np.float32)
    ref.reset_all()
    model = scenes.build_model(ops)
    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController(device=ens.hip_device)
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, w["iters"])
    ts.stepResident(model, steps)
    ts.syncToHost(model)
    xg = model.getParticles().positions()
    return bool(np.array_equal(xg.view(np.uint32), xr.view(np.uint32)))


# ---------------------------------------------------------------------------------------------------
# output: full record -> bench_detail.json, compact lines -> stdout (the headline LAST, < 4 KB)
# ---------------------------------------------------------------------------------------------------
MAX_LINE = 4096


def _r(x, sig=6):
    """Floats to `sig` significant digits (the compact lines carry measurements, not bit patterns)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_roofline(r):
    if not r:
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_kind", "achieved_traffic", "frac_traffic", "copy_ceiling_gbs", "frac_traffic_of_copy_ceiling", "frac_of_achievable", "floor_ms", "time_over_floor", "frac_valu", "valu_cycles_per_instruction_measured",
                    "valu_instructions_per_launch", "gpu_cycles_per_launch", "frac_lds", "lds_active_cycles_per_launch", "lds_bank_conflict_cycles_per_launch", "binding", "traffic", "traffic_over_compulsory", "algorithmic_bytes_per_launch",
                    "compulsory_bytes_per_launch", "avg_launch_us", "eager_launch_us", "rocprofv3_median_kernel_us", "rocprofv3_mean_kernel_us",
                    "rocprofv3_dispatches", "launches_measured"))
    out["kernel"] = str(r.get("kernel", ""))[:110]
    out["note"] = ("achieved/frac = SURVEY 8d algorithmic bytes / median device time (/ 8 TB/s; > 1: LDS-resident tiles do not move those bytes)" +
                   ("; achieved_traffic/frac_traffic = PMC HBM-side bytes per launch / the same time; frac_valu = SQ_INSTS_VALU x measured issue interval; frac_lds = LDS-array cycles, all / the launch's cycles"
                    if r.get("frac_traffic") is not None else " (no counter pass for this line)"))
    if r.get("timing_mode"):
        out["timing_mode"] = r["timing_mode"].split(":")[0]
    return out


def compact_headline(full, detail_path=None):
    """The driver-facing line: every key the bench contract names, summaries instead of tables."""
    c = full["config"]
    cfg = _pick(c, ("workload", "particles", "constraints", "colour_groups", "projections_per_substep", "rccl_ranks", "dist_backend", "oversubscribed",
                    "per_rank_ms_per_step", "rank_hip_devices", "rank_pci_bus_ids", "state_ok", "replicas_bit_identical", "shards_distinct", "replica_checksums", "shard_parity", "instances_of_rank0",
                    "device_event_ms_per_substep", "device_median_ms_per_substep", "pcie_inclusive_ms_per_step"))
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    cfg["parallelism"] = "ensemble x%d, no data-path collective" % c.get("rccl_ranks", 1)
    pv = c.get("parity_vs_reference")
    if pv:
        cfg["parity_vs_reference"] = _pick(pv, ("bit_identical", "steps", "compared_values", "max_abs"))
        cfg["parity_vs_reference"]["schedule"] = pv.get("engine_schedule")
    per = c.get("persistent") or {}
    cfg["schedule"] = {"fused": bool((c.get("plan") or {}).get("active")), "persistent": bool(per.get("active")), "folded": bool(per.get("last_folded")),
                       "refusals": per.get("refusals"), "timeouts": per.get("timeouts")}
    if detail_path:
        cfg["detail"] = detail_path
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_substep", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data"))
    out["config"] = cfg
    if full.get("roofline"):
        out["roofline"] = compact_roofline(full["roofline"])
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ms_per_substep", "single_thread", "variant", "host_logical_cpus", "nproc_threads"))
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    ex = full.get("extra_workloads") or []
    if ex:
        out["extras"] = [{"w": e.get("tag", "?"), "ms": _r(e.get("ms_per_step", e.get("ms_per_substep")), 4), "ok": bool(e.get("state_ok", False) and e.get("bit_identical") is not False),
                          "bit_identical": e.get("bit_identical")} for e in ex]      # (ok = finite + pins AND, where a reference leg ran, bit-identical; bit_identical null = no leg)
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= MAX_LINE:            # never let a long list cost the record: drop the optional parts, longest first
        for k in ("extras",):
            out.pop(k, None)
        for k in ("replica_checksums", "schedule", "instances_of_rank0", "rank_pci_bus_ids", "rank_hip_devices"):
            out["config"].pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= MAX_LINE:
        raise RuntimeError("bench.py: headline line is %d bytes (limit %d)" % (len(line), MAX_LINE))
    return line


def compact_extra(e):
    """One line per extra workload (printed BEFORE the headline)."""
    out = _pick(e, ("tag", "workload", "particles", "constraints", "colour_groups", "steps", "warmup", "ms_per_substep", "device_median_ms_per_substep", "ms_per_step", "projections_per_s", "state_ok",
                    "host_scene_build_s", "bit_identical", "compared_values", "parity_steps", "contacts_deformable_total", "contacts_floor_total", "contacts_max", "sub_steps", "iterations",
                    "reference_ms_per_step", "reference_reproduces_fixture", "same_bits_as_default_build", "error"))
    if "workload" in out:
        out["workload"] = str(out["workload"])[:220]
    if e.get("roofline"):
        out["roofline"] = compact_roofline(e["roofline"])
    per = e.get("persistent") or {}
    if per:
        out["schedule"] = {"fused": bool((e.get("plan") or {}).get("active")), "persistent": bool(per.get("active"))}
    line = json.dumps({"extra": _r(out)}, separators=(",", ":"))
    return line if len(line) < MAX_LINE else json.dumps({"extra": _r(_pick(out, ("tag", "ms_per_substep", "state_ok")))}, separators=(",", ":"))


def emit(full, write_detail=True):
    """Full record to the side file, compact lines to stdout, headline last."""
    detail_path = None
    if write_detail:
        for path in [os.environ.get("PBDX_BENCH_DETAIL"), os.path.join(ROOT, "bench_detail.json")]:
            if not path:
                continue
            try:
                with open(path, "w") as fh:
                    json.dump(full, fh)
                detail_path = detail_path or os.path.relpath(path, ROOT)
            except OSError:
                pass
    for e in full.get("extra_workloads") or []:
        print(compact_extra(e), flush=True)
    print(compact_headline(full, detail_path), flush=True)


def dry_line(path):
    """No GPU: the compact lines of a stored full record, stretched to the worst case the driver will see (8 ranks)."""
    with open(path) as fh:
        full = json.load(fh)
    c = full["config"]
    c["rccl_ranks"] = 8
    c["per_rank_ms_per_step"] = [(c.get("per_rank_ms_per_step") or [1.0])[0] * (1 + 1e-3 * i) for i in range(8)]
    c["replica_checksums"] = [(c.get("replica_checksums") or ["0" * 16])[0]] * 8
    for i, e in enumerate(full.get("extra_workloads") or []):
        e.setdefault("tag", "extra%d" % i)
    emit(full, write_detail=False)


# ---------------------------------------------------------------------------------------------------
# N ranks from one command
# ---------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what torch.distributed.run provides), pass
    rank 0's JSON line through, fail if any rank fails."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PBDX_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while ALL ranks are watched: a rank that dies (no device, HIP error) must end the
    # job with its exit code -- never leave the others waiting at a rendezvous or a barrier
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:               # exactly the processes started above
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = (chunks[0] if chunks else b"").decode()
    for line in out0.splitlines():      # the JSON lines go to stdout (headline last); library chatter of the rank (gloo / RCCL banners) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit("bench.py: rank %d exited with code %d; the other ranks were stopped (exit codes %r)" % (failed[0], failed[1], rcs))
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %r" % (rcs,))


def single_process_ensemble(args):
    """BASELINE configs[3] over N devices of ONE process through the C ABI (include/pbdx.h pbdx_ensemble_*): --instances 200x200 sheets per device in one
    instanced model, split into contiguous blocks, one engine + stream + host thread per device, no torch.distributed.  Same contract line as the
    one-process-per-GPU form (weak scaling: value = all devices' projections / wall time of the K timed steps)."""
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd import scenes
    ndev = pbd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if max(devices) >= ndev:
        raise SystemExit("bench.py --single-process: device list %r but %d GPU(s) visible" % (devices, ndev))
    size = args.size or 200
    total = args.instances * len(devices)
    ops = scenes.cloth_spec(size, size, 4, 3, instances=total, instance_offset=(0.0, 0.0, 12.0), instanced=True)
    t0 = time.perf_counter()
    model = scenes.build_model(ops)
    model.initConstraintGroups()
    t_build = time.perf_counter() - t0
    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    e = pbd.DeviceEnsemble(devices)
    e.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    e.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, args.iters)
    e.setModel(model)
    e.step(max(args.warmup, 1))
    t0 = time.perf_counter()
    e.step(args.steps)                     # returns when every device's stream has drained
    t = time.perf_counter() - t0
    e.gather()
    x = model.getParticles().positions()
    ok = bool(np.all(np.isfinite(x)))
    nc = model.numConstraints()
    blocks = [e.shard(i) for i in range(e.numShards())]
    out = {"metric": "constraint-projections/s", "value": nc * args.iters * args.steps / t, "unit": "projections/s", "n_gpus": len(set(devices)), "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * t / args.steps, "ms_per_substep": 1e3 * t / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3] in ONE process (pbdx_ensemble_*): %d x (%dx%d) cloth instances, %d per engine, XPBD distance + XPBD isometric bending, %d iterations x 1 substep" % (
                          total, size, size, args.instances, args.iters),
                      "parallelism": "single process, %d engines on devices %r, contiguous blocks of instances, no collective" % (len(devices), devices),
                      "particles": model.getParticles().size(), "constraints": nc, "projections_per_substep": nc * args.iters, "state_ok": ok,
                      "blocks": blocks, "per_block_ms_per_step": [b["last_step_ms"] / args.steps for b in blocks], "host_scene_build_s": t_build,
                      "engines": [e.shardSolver(i).describe()[-150:] for i in range(e.numShards()) if blocks[i]["end"] > blocks[i]["begin"]]}}
    print(json.dumps(_r(out), separators=(",", ":")), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2", help="c5: only the configs[4]-shaped contact scene lines (8 and 5 substeps), no headline")
    ap.add_argument("--size", type=int, default=None, help="cloth is size x size particles (default 1000; 200 for c4)")
    ap.add_argument("--instances", type=int, default=64, help="c4 (weak scaling): cloth instances per GPU; c3 --bars: bars per GPU")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="c4: fixed instances per GPU (weak) or --total-instances sharded over the GPUs (strong)")
    ap.add_argument("--total-instances", type=int, default=512, help="c4 --scaling strong: instances of the whole job (BASELINE configs[3]: 512)")
    ap.add_argument("--bars", action="store_true", help="c3: batch --instances independent bars per GPU (the single bar is latency-bound by construction)")
    ap.add_argument("--solid-method", type=int, default=2, help="c3: addSolidConstraints method (2 FEM tet, 4 strain tet, 6 XPBD distance+volume)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nproc", action="store_true", help="cpu_baseline: also ONE reference step at OMP_NUM_THREADS = nproc in a child process (a minute on a 256-CPU host; off by default)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra_workloads (configs[2] variants, configs[3] block)")
    ap.add_argument("--xcd-remap", type=int, default=None)
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph")
    ap.add_argument("--fuse", type=int, default=None, help="1 = colour-fused LDS tiles (default), 0 = one launch per colour")
    ap.add_argument("--tile", type=int, default=None, help="particles owned by one tile (0 = auto)")
    ap.add_argument("--fuse-block", type=int, default=None)
    ap.add_argument("--max-seg", type=int, default=None, help="max colours fused into one launch")
    ap.add_argument("--lds-particles", type=int, default=None)
    ap.add_argument("--persistent", type=int, default=None, help="PBDX_OPT_PERSISTENT: 1 = one launch per substep where measured faster (default), 0 = one launch per segment, 2 = always")
    ap.add_argument("--wgs-per-cu", type=int, default=None, help="PBDX_OPT_PERSISTENT_WGS_PER_CU: tiles resident per CU in the one-launch schedule")
    ap.add_argument("--contacts", action="store_true", help="also time the step with two static colliders (contact detection + velocity solve per step)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than GPUs (ranks share devices; smoke test of the N>1 path, not a measurement)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default=None, help="torch.distributed backend (default: nccl = RCCL when GPUs are visible)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-steps", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dry-line", nargs="?", const=os.path.join(ROOT, "profiles", "r02_bench.json"), default=None,
                    help="no GPU: print the compact lines for a stored FULL record (default profiles/r02_bench.json)")
    ap.add_argument("--check-shards", action="store_true", help="c4: every rank also steps ITS instances with the reference (oracle/_ref f32) and compares bits (small sizes only)")
    ap.add_argument("--single-process", action="store_true", help="c4 with --gpus N in ONE process: the C ABI's pbdx_ensemble_* (one engine per device, all devices stepping at once, no torch.distributed); "
                    "a second way to measure ensemble scaling")
    ap.add_argument("--devices", type=str, default=None, help="--single-process: comma-separated HIP device list (default 0..N-1; a device may repeat: smoke test on one GPU)")
    ap.add_argument("--rank-selftest", action="store_true", help="launcher self-test: the ranks rendezvous, exchange their rank numbers and exit (no GPU work, no metric)")
    args = ap.parse_args()

    if args.dry_line:
        return dry_line(args.dry_line)
    if args.single_process:
        return single_process_ensemble(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        return spawn_ranks(args.gpus, sys.argv[1:])

    if args.rank_selftest:
        from positionbaseddynamics_amd.ensemble import Ensemble
        if os.environ.get("PBDX_SELFTEST_FAIL_RANK") == os.environ.get("RANK", "0"):      # test hook: this rank dies before the rendezvous
            raise SystemExit(3)
        # (--dist-backend nccl at one rank = the RCCL path on a one-GPU box: init with device_id, all-reduces, barrier, destroy)
        ens = Ensemble(backend=args.dist_backend or os.environ.get("PBDX_DIST_BACKEND") or "gloo", force_init=True)
        sys.stderr.write("[bench rank %d/%d] %r\n" % (ens.rank, ens.world, ens.describe_device()))
        ranks = ens.gather_floats(ens.rank)
        t = ens.max_time(0.5 + ens.rank)
        rank_sum = ens.sum_count(ens.rank + 1)
        ens.barrier()
        if ens.rank == 0:
            print(json.dumps({"selftest": "ranks", "n_gpus": None, "rccl_ranks": ens.world, "dist_backend": ens.backend, "ranks": ranks, "sum": rank_sum, "max_time": t, "spawned_by_bench": os.environ.get("PBDX_SPAWNED") == "1"}), flush=True)
        ens.close()
        return

    if args.size is None:
        args.size = 200 if args.workload == "c4" else 1000
    import torch
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd.ensemble import Ensemble
    ndev = pbd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > ndev and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible; one process per GPU (use --oversubscribe only to smoke-test the N>1 path)" % (world_env, ndev))
    local_env = int(os.environ.get("LOCAL_RANK", "0"))
    if world_env > 1 and local_env >= ndev and not args.oversubscribe:
        raise SystemExit("bench.py: rank %s has LOCAL_RANK %d but only %d GPU(s) are visible to it (HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r)" % (
            os.environ.get("RANK"), local_env, ndev, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
    # (an explicitly requested backend is initialised even at world size 1: the RCCL path on a one-GPU box)
    ens = Ensemble(oversubscribe=args.oversubscribe, num_devices=ndev, backend=args.dist_backend, force_init=args.dist_backend is not None)
    rank, world = ens.rank, ens.world
    if world == 1:
        torch.cuda.set_device(0)
    # which GPU every rank chose (stderr, one line per rank) + a check on rank 0 that no two ranks share a physical device
    dev = ens.describe_device()
    sys.stderr.write("[bench rank %d/%d] LOCAL_RANK %d -> HIP device %d of %s visible (%s, pci %s); backend %s; HIP_VISIBLE_DEVICES=%r ROCR_VISIBLE_DEVICES=%r CUDA_VISIBLE_DEVICES=%r\n" % (
        rank, world, dev["local_rank"], dev["hip_device"], dev["visible_devices"], dev["name"], "%04x" % dev["pci_bus_id"] if dev["pci_bus_id"] is not None else "?",
        dev["backend"], dev["HIP_VISIBLE_DEVICES"], dev["ROCR_VISIBLE_DEVICES"], dev["CUDA_VISIBLE_DEVICES"]))
    sys.stderr.flush()
    rank_devices = ens.gather_ints(dev["hip_device"])
    rank_pci = ens.gather_ints(dev["pci_bus_id"] if dev["pci_bus_id"] is not None else -1)
    if world > 1 and not args.oversubscribe:
        keys = list(zip(rank_devices, rank_pci))
        if len(set(keys)) != world:
            raise SystemExit("bench.py: two ranks chose the same GPU (HIP device, pci bus) per rank = %r: one process per GPU" % (keys,))

    if args.workload == "c5":
        for sub in (8, 5):
            print(compact_extra(run_c5(ens, sub)), flush=True)
        ens.close()
        return
    if args.pmc_child:
        from positionbaseddynamics_amd import _ffi
        for mode in range(4):
            _ffi.check(_ffi.lib.pbdx_debug_stream(ens.hip_device, CALIB_BYTES, mode), "pbdx_debug_stream")
        args.no_roofline = args.no_cpu_baseline = args.no_extras = True
        args.steps, args.warmup = (args.child_steps, 10) if args.child_steps else (2, 1)

    w = {"workload": args.workload, "size": args.size, "instances": args.instances, "bars": args.bars, "solid_method": args.solid_method,
         "iters": args.iters, "scaling": args.scaling, "total_instances": args.total_instances}
    opts = {"xcd_remap": args.xcd_remap, "block": args.block, "fuse": args.fuse, "tile": args.tile, "fuse_block": args.fuse_block,
            "max_seg": args.max_seg, "lds_particles": args.lds_particles, "persistent": args.persistent, "no_graph": args.no_graph, "wgs_per_cu": args.wgs_per_cu, "cpu_nproc": args.cpu_nproc}

    res = run_workload(w, opts, ens, args.steps, args.warmup, with_roofline=not args.no_roofline, with_traffic=not args.no_traffic and not args.pmc_child,
                       with_pcie=not args.pmc_child, with_contacts=args.contacts)
    t_max = ens.max_time(res["t_local"])                               # max over ranks
    total_constraints = ens.sum_count(res["n_constraints"])            # all ranks
    per_rank_ms = ens.gather_floats(1e3 * res["t_local"] / args.steps)
    # cross-GPU parity: ranks that simulate the same scene (c2, c3: replicas; c4: identical instance blocks when the blocks
    # have equal size) must hold the same bits -- one tiny all-reduce of per-rank checksums
    sums = ens.gather_checksums([res["checksum"]], world)
    # c2 / c3: every rank holds the same scene (replicas must agree bit for bit); c4: ranks hold DIFFERENT instances
    # (global instances begin .. end-1 with their own translations), so equal checksums would be a sharding bug
    replicas_identical = (len(set(sums)) == 1) if args.workload != "c4" else None
    shards_distinct = (len(set(sums)) == world) if args.workload == "c4" else None
    shard_parity = None
    if args.check_shards and args.workload == "c4":
        ok_ranks = ens.sum_count(1 if shard_vs_reference(w, ens, opts, 3) else 0)
        shard_parity = {"ranks_checked": world, "ranks_bit_identical_to_reference": ok_ranks, "steps": 3}
    all_ok = ens.sum_count(1 if res["state_ok"] else 0) == world

    value = total_constraints * args.iters * args.steps / t_max
    ms_per_step = 1e3 * t_max / args.steps
    stats = res["stats"]
    n_physical = min(world, ndev) if args.oversubscribe else world
    out = {
        "metric": "constraint-projections/s", "value": value, "unit": "projections/s",
        "n_gpus": n_physical, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_substep": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling if args.workload == "c4" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": res["desc"],
                   "particles": res["n_particles"], "constraints": res["n_constraints"], "colour_groups": res["n_groups"],
                   "constraints_all_ranks": total_constraints,
                   "projections_per_substep": res["n_constraints"] * args.iters,
                   "parallelism": "ensemble x%d (independent instances per GPU, no cross-GPU constraints, no data-path collective)" % world,
                   "rccl_ranks": world, "dist_backend": ens.backend, "oversubscribed": bool(args.oversubscribe and world > ndev),
                   "rank_hip_devices": rank_devices, "rank_pci_bus_ids": ["%04x" % b if b >= 0 else None for b in rank_pci],
                   "per_rank_ms_per_step": per_rank_ms, "replica_checksums": ["%016x" % c for c in sums],
                   "state_ok": all_ok, "replicas_bit_identical": replicas_identical, "shards_distinct": shards_distinct, "shard_parity": shard_parity,
                   "instances_of_rank0": list(ens.shard(args.total_instances if args.scaling == "strong" else args.instances * world)) if args.workload == "c4" else None,
                   "state_checksum": "%016x" % sums[0],
                   "device_event_ms_per_substep": stats["total_ms"] / max(args.steps, 1),
                   "device_median_ms_per_substep": (res.get("substep_device") or {}).get("median_ms"), "substep_device_times": res.get("substep_device"),
                   "algorithmic_GB_per_substep": stats["algorithmic_bytes"] / max(args.steps, 1) / 1e9,
                   "whole_substep_algorithmic_GBs": stats["algorithmic_bytes"] / max(stats["total_ms"], 1e-9) / 1e6,
                   "host_scene_build_s": res["t_build"], "contacts": res.get("contacts"),
                   "pcie_inclusive_ms_per_step": res.get("pcie_ms"), "pcie_inclusive_pinned_ms_per_step": res.get("pcie_pinned_ms"),
                   "plan": res["plan"], "persistent": res["persistent"], "engine": res["engine"]},
    }
    if res.get("roofline"):
        out["roofline"] = res["roofline"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        size = args.size if args.workload == "c2" else 400
        cb, parity = cpu_baseline(size, args.iters, opts, ens.hip_device)
        out["cpu_baseline"] = cb
        out["config"]["parity_vs_reference"] = parity

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2" and not args.pmc_child:
        # the other BASELINE workloads that fit one GPU, witnessed in the same line (shorter runs; same engine, defaults)
        extras = []
        base = {"size": 200, "instances": 64, "bars": False, "solid_method": 2, "iters": args.iters, "scaling": "weak", "total_instances": 512}
        # (tag, workload, PMC traffic passes, warm-up steps).  SURVEY 8d: timing after >= 20 warm-up steps.  The E = 1 bar of configs[2]
        # collapses under gravity and more of its tets take the inversion-handling branch as the run goes on: `c3_fem_tets_late` times the same
        # substep after 100 warm-up steps; `c3_x32_fem` is the form that fills the GPU with this workload (32 independent bars, instanced)
        for tag, ew, want_traffic, wu in (("c3_fem_tets", {**base, "workload": "c3", "solid_method": 2}, True, 20),
                                          ("c3_fem_tets_late", {**base, "workload": "c3", "solid_method": 2}, False, 100),
                                          ("c3_strain_tets", {**base, "workload": "c3", "solid_method": 4}, False, 20),
                                          ("c3_xpbd_distance_volume", {**base, "workload": "c3", "solid_method": 6}, False, 20),
                                          ("c3_x32_fem", {**base, "workload": "c3", "solid_method": 2, "bars": True, "instances": 32}, False, 20),
                                          ("c4_block_64x200x200", {**base, "workload": "c4"}, True, 20)):
            nsteps = max(10, min(args.steps, 30))
            try:
                r = run_workload(ew, {}, ens, nsteps, wu, with_roofline=not args.no_roofline, with_traffic=want_traffic and not args.no_traffic)
            except Exception as e:  # an extra must never cost the headline line
                extras.append({"tag": tag, "workload": ew["workload"], "error": repr(e)})
                continue
            ms = 1e3 * r["t_local"] / nsteps
            # every extra line carries a reference leg of its own; the late-state bar's runs INTO the late state (warm-up + 2 steps from rest: crushed
            # tets in the inversion branch on both sides; 102 steps of the reference at up to 32 threads take a few seconds)
            par = None if args.no_cpu_baseline else extra_parity(ew, ens, ens.hip_device, steps=(wu + 2) if tag == "c3_fem_tets_late" else 2)
            extras.append({"tag": tag, "workload": r["desc"], "particles": r["n_particles"], "constraints": r["n_constraints"], "colour_groups": r["n_groups"],
                           "parity_vs_reference": par, "bit_identical": (par or {}).get("bit_identical"),
                           "steps": nsteps, "warmup": wu, "ms_per_substep": ms, "device_median_ms_per_substep": (r.get("substep_device") or {}).get("median_ms"),
                           "projections_per_s": r["n_constraints"] * ew["iters"] * nsteps / r["t_local"],
                           "state_ok": r["state_ok"], "host_scene_build_s": r["t_build"], "plan": r["plan"], "persistent": r["persistent"],
                           "engine": r["engine"], "roofline": r.get("roofline")})
        # the opt-in contracted build (csrc/Makefile: libpbdx_fma.so, -ffp-contract=fast): same workload, same command, in a child process that
        # loads that library.  NOT bit-identical to the float reference (tests: inside the fp32 envelope around f64); reported beside the headline,
        # never as the headline
        fma_lib = os.path.join(ROOT, "positionbaseddynamics_amd", "_lib", "libpbdx_fma.so")
        if os.path.exists(fma_lib) and not os.environ.get("PBDX_LIB"):
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-traffic", "--no-extras", "--no-roofline", "--steps", str(args.steps),
                                     "--warmup", str(args.warmup), "--iters", str(args.iters), "--size", str(args.size)], env=dict(os.environ, PBDX_LIB=fma_lib, PBDX_BENCH_DETAIL="/dev/null"),
                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
                cd = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
                extras.append({"tag": "c2_fma_contract_build", "workload": "the headline workload on the OPT-IN library built with -ffp-contract=fast (not bit-identical to the float reference; inside the fp32 envelope)",
                               "steps": cd["steps"], "warmup": cd["warmup"], "ms_per_substep": cd["ms_per_substep"], "device_median_ms_per_substep": cd["config"].get("device_median_ms_per_substep"),
                               "projections_per_s": cd["value"], "state_ok": cd["config"]["state_ok"],
                               "same_bits_as_default_build": cd["config"]["replica_checksums"] == out["config"]["replica_checksums"]})
            except Exception as e:
                extras.append({"tag": "c2_fma_contract_build", "error": repr(e)})
        for sub in (8, 5):
            try:
                extras.append(run_c5(ens, sub))
            except Exception as e:
                extras.append({"tag": "c5_armadillo_%dsub" % sub, "error": repr(e)})
        out["extra_workloads"] = extras

    if rank == 0:
        emit(out, write_detail=not args.pmc_child)
    ens.close()


if __name__ == "__main__":
    main()
