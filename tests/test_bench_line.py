"""bench.py's driver-facing output: the LAST stdout line is the compact headline record and must fit the driver's 8 KB
stdout tail with room to spare (< 4 KB) whatever the number of ranks; the extra workloads are separate, earlier lines.
Round 2's record was lost because ONE line had grown to 20 KB (task 1).  `--dry-line` replays a stored FULL
record (profiles/r02_bench.json: the 20 KB line itself) through the same formatting code without a GPU."""
import json
import os
import subprocess
import sys

from tests import util

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")
CONFIG_KEYS = ("workload", "particles", "constraints", "rccl_ranks", "per_rank_ms_per_step", "parity_vs_reference")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "algorithmic_bytes_per_launch",
                 "compulsory_bytes_per_launch", "avg_launch_us")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "ms_per_substep", "single_thread")


def _dry(path=None):
    cmd = [sys.executable, os.path.join(util.ROOT, "bench.py"), "--dry-line"] + ([path] if path else [])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    return [ln for ln in p.stdout.splitlines() if ln.strip()]


def test_headline_is_the_last_line_and_fits():
    lines = _dry()
    assert all(len(ln) < 4096 for ln in lines), [len(ln) for ln in lines]
    assert sum(len(ln) + 1 for ln in lines) < 8192          # headline AND extras inside the driver's tail
    head = json.loads(lines[-1])
    for k in HEADLINE_KEYS:
        assert k in head, k
    for k in CONFIG_KEYS:
        assert k in head["config"], k
    for k in ROOFLINE_KEYS:
        assert k in head["roofline"], k
    for k in CPU_KEYS:
        assert k in head["cpu_baseline"], k
    assert len(head["config"]["per_rank_ms_per_step"]) == 8          # --dry-line stretches the record to the 8-rank case
    pv = head["config"]["parity_vs_reference"]
    assert set(("bit_identical", "steps", "compared_values")) <= set(pv)
    assert head["roofline"]["frac"] == head["roofline"]["achieved"] / head["roofline"]["peak"] or \
        abs(head["roofline"]["frac"] - head["roofline"]["achieved"] / head["roofline"]["peak"]) < 1e-4
    extras = [json.loads(ln) for ln in lines[:-1]]
    assert extras and all("extra" in e and "ms_per_substep" in e["extra"] for e in extras)


def test_oversized_optional_parts_are_dropped_not_fatal(tmp_path):
    """A pathological record (very long strings everywhere) still yields a parseable headline under the limit."""
    with open(os.path.join(util.ROOT, "profiles", "r02_bench.json")) as fh:
        full = json.load(fh)
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["extra_workloads"] = [dict(e, workload="x" * 3000, tag="t%d" % i) for i, e in enumerate(full["extra_workloads"] * 10)]
    path = tmp_path / "full.json"
    path.write_text(json.dumps(full))
    lines = _dry(str(path))
    assert all(len(ln) < 4096 for ln in lines)
    head = json.loads(lines[-1])
    assert head["metric"] == "constraint-projections/s" and head["roofline"]["frac"] > 0 and head["cpu_baseline"]["cores"] >= 1
