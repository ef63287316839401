"""Ensemble sharding of independent scene instances over GPUs.

A single sheet / bar is one connected colour-sequential Gauss-Seidel problem and does not shard;
what shards is a set of independent instances.  One process per GPU (`torch.distributed`; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for the tests), contiguous blocks of instances per rank, NO
collective on the data path: RCCL/gloo carry only the barrier, the max-over-ranks time, a few
counters and (for parity checks) per-instance checksums."""
import os

import numpy as np


def shard_range(total, world, rank):
    """Contiguous block [begin, end) of `total` instances owned by `rank` (sizes differ by at most 1): include/pbdx.h
    pbdx_ensemble_shard, the one definition C++ hosts and this helper share."""
    import ctypes as C
    from . import _ffi
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    b, e = C.c_uint64(0), C.c_uint64(0)
    _ffi.check(_ffi.lib.pbdx_ensemble_shard(int(total), int(world), int(rank), C.byref(b), C.byref(e)), "pbdx_ensemble_shard")
    return int(b.value), int(e.value)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def checksum(positions):
    """Order-sensitive 64-bit checksum of a float32 position array (bit-exact comparisons across ranks)."""
    a = np.ascontiguousarray(positions, dtype=np.float32).view(np.uint32).astype(np.uint64).reshape(-1)
    w = (np.arange(a.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))
    with np.errstate(over="ignore"):
        return int(np.bitwise_xor.reduce(a * w + (a << np.uint64(17)))) & 0x7FFFFFFFFFFFFFFF


class Ensemble:
    """Process-group plumbing shared by bench.py and the tests."""

    def __init__(self, backend=None, oversubscribe=False, num_devices=None, force_init=False):
        """oversubscribe: more ranks than GPUs are allowed to share devices (gloo backend, HIP device = local_rank mod
        num_devices): a smoke test of the N>1 code path on a box with fewer GPUs, never a measurement.
        force_init: create the process group even at world size 1 (the GPU test of the RCCL path on a one-GPU box)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        self.hip_device = self.local_rank if self.world > 1 else 0
        if oversubscribe and num_devices and self.world > num_devices:
            backend = backend or "gloo"
            self.hip_device = self.local_rank % num_devices
        if self.world > 1 or force_init:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if force_init and self.world == 1:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            # PBDX_DIST_BACKEND: smoke-testing the N>1 path on a box with fewer GPUs than ranks (gloo)
            backend = backend or os.environ.get("PBDX_DIST_BACKEND") or None
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                # one process per GPU: LOCAL_RANK is the HIP device index among the VISIBLE devices.  Checked before the
                # rendezvous: a rank without a device must fail with a message, not leave the others waiting in init.
                nvis = torch.cuda.device_count() if torch.cuda.is_available() else 0
                if self.local_rank >= nvis:
                    raise RuntimeError("Ensemble: rank %d has LOCAL_RANK %d but only %d HIP device(s) are visible (HIP_VISIBLE_DEVICES=%r, "
                                       "ROCR_VISIBLE_DEVICES=%r): one process per GPU" % (self.rank, self.local_rank, nvis,
                                       os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
                self.hip_device = self.local_rank
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
                # ranks may share a GPU in this mode: PBDX_DEVICE_OVERRIDE pins the HIP device index
                if os.environ.get("PBDX_DEVICE_OVERRIDE") is not None:
                    self.local_rank = int(os.environ["PBDX_DEVICE_OVERRIDE"])
                    self.hip_device = self.local_rank
            self.dist = dist
        self.backend = backend

    def describe_device(self):
        """What this rank runs on: (rank, LOCAL_RANK) -> HIP device index among the visible devices, its name and PCI bus id, and
        the visibility variables in force.  bench.py prints it per rank and gathers the PCI ids so that rank 0 can check that no
        two ranks share a physical GPU."""
        info = {"rank": self.rank, "world": self.world, "local_rank": self.local_rank, "hip_device": self.hip_device, "backend": self.backend,
                "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
                "CUDA_VISIBLE_DEVICES": os.environ.get("CUDA_VISIBLE_DEVICES"), "visible_devices": None, "name": None, "pci_bus_id": None}
        try:
            import torch
            if torch.cuda.is_available():
                info["visible_devices"] = torch.cuda.device_count()
                prop = torch.cuda.get_device_properties(self.hip_device)
                info["name"] = prop.name
                bus = getattr(prop, "pci_bus_id", None)
                dom = getattr(prop, "pci_domain_id", 0) or 0
                if bus is not None:
                    info["pci_bus_id"] = (int(dom) << 8) | int(bus)
        except Exception as e:  # a description must never cost the run
            info["error"] = repr(e)
        return info

    def gather_ints(self, value):
        """One integer per rank as the list over ranks on every rank (SUM all-reduce of a one-hot vector)."""
        if self.dist is None:
            return [int(value)]
        import torch
        v = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        v[self.rank] = int(value)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        return [int(x) for x in v.cpu().tolist()]

    def shard(self, total):
        return shard_range(total, self.world, self.rank)

    def _sync_device(self):
        if self.device is not None and self.device.type == "cuda":
            import torch
            torch.cuda.synchronize()

    def barrier(self):
        self._sync_device()
        if self.dist is not None:
            self.dist.barrier()
        self._sync_device()

    def _reduce(self, value, op_name, dtype):
        if self.dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=dtype, device=self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op_name))
        return t.item()

    def max_time(self, seconds):
        import torch
        return float(self._reduce(float(seconds), "MAX", torch.float64))

    def sum_count(self, n):
        import torch
        return int(self._reduce(int(n), "SUM", torch.int64))

    def gather_floats(self, value):
        """One float per rank, returned as the list over ranks on every rank (SUM all-reduce of a one-hot vector)."""
        if self.dist is None:
            return [float(value)]
        import torch
        v = torch.zeros(self.world, dtype=torch.float64, device=self.device)
        v[self.rank] = float(value)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in v.cpu().tolist()]

    def gather_checksums(self, local, total):
        """All ranks contribute the checksums of their instances; returns the full list (length `total`)
        on every rank.  Implemented as a SUM all-reduce of a zero-padded vector (instances are disjoint)."""
        begin, end = self.shard(total)
        if len(local) != end - begin:
            raise ValueError("rank %d owns %d instances, got %d checksums" % (self.rank, end - begin, len(local)))
        if self.dist is None:
            return list(local)
        import torch
        v = torch.zeros(total, dtype=torch.int64, device=self.device)
        v[begin:end] = torch.tensor([int(c) for c in local], dtype=torch.int64)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        return [int(x) for x in v.cpu().tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
