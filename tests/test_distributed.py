"""The N>1 path of bench.py on CPU: world_size-2 gloo process group, contiguous instance sharding,
barrier, max-over-ranks time, counter sums and the per-instance checksum exchange used for ensemble
parity.  There is no collective on the data path (instances are independent), so this IS the whole
multi-GPU machinery; the GPU kernels are covered by the -m gpu tests."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import util

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from positionbaseddynamics_amd.ensemble import Ensemble, checksum
from oracle import port
from tests import util

ens = Ensemble(backend="gloo")
total = 5
begin, end = ens.shard(total)
# every instance: a tiny cloth stepped by the plain-C oracle port (CPU stand-in for the GPU engine in this
# plumbing test); instance k is pinned differently so that checksums differ per instance
sums = []
proj = 0
for k in range(begin, end):
    o = port.Port("f32")
    spec = util.cloth_spec(6 + k, 5, 4, 3)
    util.apply_ref(o, spec)
    o.set_time_step_size(0.005); o.set_params(1, 3, 0); o.step(2)
    sums.append(checksum(o.positions().astype(np.float32)))
    proj += o.num_constraints() * 3 * 2
ens.barrier()
t_max = ens.max_time(0.25 * (ens.rank + 1))
total_proj = ens.sum_count(proj)
allsums = ens.gather_checksums(sums, total)
if ens.rank == 0:
    print(json.dumps({"t_max": t_max, "total_proj": total_proj, "sums": allsums, "world": ens.world}))
ens.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    from positionbaseddynamics_amd.ensemble import shard_range
    for total in (0, 1, 5, 8, 512, 513):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, e = shard_range(total, world, r)
                assert 0 <= b <= e <= total
                got += list(range(b, e))
            assert got == list(range(total))
            sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo_ensemble(tmp_path):
    import json
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": util.ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2
    assert res["t_max"] == pytest.approx(0.5)                      # max over ranks, not the local value
    # single-process ground truth over all 5 instances
    from oracle import port as oport
    from positionbaseddynamics_amd.ensemble import checksum
    want, proj = [], 0
    for k in range(5):
        o = oport.Port("f32")
        util.apply_ref(o, util.cloth_spec(6 + k, 5, 4, 3))
        o.set_time_step_size(0.005); o.set_params(1, 3, 0); o.step(2)
        want.append(checksum(o.positions().astype(np.float32)))
        proj += o.num_constraints() * 3 * 2
    assert res["sums"] == want                                      # every instance exactly once, bit-exact
    assert res["total_proj"] == proj
    assert len(set(want)) == 5


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run would), rendezvous and print ONE line from rank 0.  The launcher self-test stops
    after the rendezvous (no GPU here); the same spawner carries the real run on a GPU box (profiles/: bench --gpus 2)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--rank-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["rccl_ranks"] == 2 and res["ranks"] == [0.0, 1.0] and res["max_time"] == pytest.approx(1.5) and res["spawned_by_bench"]
    # under a launcher (WORLD_SIZE set) the script must NOT spawn again
    port = _free_port()
    procs = []
    for rank in range(2):
        e2 = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--rank-selftest"], env=e2,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [q.communicate(timeout=300) for q in procs]
    assert all(q.returncode == 0 for q in procs), outs[0][1][-2000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["rccl_ranks"] == 2 and not res["spawned_by_bench"] and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
