"""Self-validation of the first N > 1 launch (frame-parallel Stage-3).

The reference trusts its launcher (/root/reference/lab4d/train.py:28-36 sets the device from LOCAL_RANK and calls
init_process_group; lab4d/utils/gpu_utils.py:6-128 hands GPUs to workers) and finds out about a bad rendezvous from a hang.
No multi-GPU box is reachable from the build container, so the first real 8-GPU run of this code is the driver's: it carries
its own evidence.  `collective_self_check` runs right after init_process_group and returns a small dict that goes on the
bench's JSON line (key "rccl") and into train.py's first log line:

  world / backend / ranks_seen (all-reduce of 1) / rank_sum_ok (all-reduce of the rank ids) / device_of_each_rank /
  payload_bytes / allreduce_ms_p50, _min, _max over `reps` standalone all-reduces of the step's payload, each bracketed by
  a device synchronise / frames_of_each_rank_head (all-gather of the first frame ids every rank renders).

It aborts (RuntimeError on every rank) when two ranks share a PHYSICAL device under RCCL -- identified by (host name, PCI
domain / bus / device), so that a multi-node launch (local device indices repeat across nodes) and per-process
HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES isolation (every rank sees its GPU as device 0) both pass -- or when a sum
comes back wrong, before any timing is taken.  A rank whose device index is not its LOCAL_RANK although it sees several
devices is reported in the result (`device_is_local_rank: false`) and warned about, not aborted: the launcher may have
mapped devices on purpose.
"""
from __future__ import annotations

import socket
import statistics
import time
import warnings
import zlib

import torch


def shared_physical_devices(identities) -> list:
    """The (host, pci domain, bus, device) keys that more than one rank reported; the local index (last entry) is not part
    of the key: two nodes x eight GPUs repeat every index, and isolated ranks all report index 0."""
    seen, shared = set(), []
    for i in identities:
        key = tuple(i[:4])
        if key in seen and key not in shared:
            shared.append(key)
        seen.add(key)
    return shared


def physical_device_id(device) -> list:
    """[host, pci domain, pci bus, pci device, local index] of a rank's device: unique per physical GPU across nodes and under
    visible-device isolation.  CPU ranks (gloo tests) and builds without PCI properties fall back to (host, -1, -1, index)."""
    host = zlib.crc32(socket.gethostname().encode())
    if device.type != "cuda":
        return [host, -1, -1, -1, -1]
    idx = device.index if device.index is not None else torch.cuda.current_device()
    prop = torch.cuda.get_device_properties(idx)
    dom, bus, dev = (getattr(prop, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if bus is None:
        return [host, -1, -1, idx, idx]
    return [host, int(dom or 0), int(bus), int(dev or 0), idx]


def collective_self_check(dist, device, local_rank: int, payload: torch.Tensor, frames_head, backend: str = "nccl",
                          reps: int = 20) -> dict:
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = device.type == "cuda"
    local_ok = True
    if on_gpu and backend != "gloo":
        cur = torch.cuda.current_device()
        # (only meaningful when the process sees more than one device: under per-process visible-device isolation every
        # rank's GPU is its device 0)
        if torch.cuda.device_count() > 1 and (cur != local_rank or device.index != local_rank):
            local_ok = False
            warnings.warn(f"rank {rank}: HIP device {cur} (tensor device {device}) is not LOCAL_RANK {local_rank}; continuing -- "
                          "the physical-device check below decides")
    one = torch.ones(1, device=device, dtype=torch.float64)
    dist.all_reduce(one)
    ids = torch.tensor([float(rank)], device=device, dtype=torch.float64)
    dist.all_reduce(ids)
    devs = [torch.zeros(5, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(devs, torch.tensor(physical_device_id(device), dtype=torch.int64, device=device))
    head = list(frames_head)[:4] + [-1] * max(0, 4 - len(list(frames_head)[:4]))
    heads = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(heads, torch.tensor(head, dtype=torch.int64, device=device))
    ranks_seen, rank_sum = int(round(one.item())), int(round(ids.item()))
    identities = [tuple(int(x) for x in d.tolist()) for d in devs]   # (host, pci domain, bus, device, local index)
    devices = [i[4] for i in identities]
    out = {"world": world, "backend": "rccl" if backend == "nccl" else backend, "ranks_seen": ranks_seen,
           "rank_sum_ok": rank_sum == world * (world - 1) // 2, "device_of_each_rank": devices,
           "hosts": len({i[0] for i in identities}), "device_is_local_rank": local_ok,
           "payload_bytes": int(payload.numel() * payload.element_size()),
           "frames_of_each_rank_head": [[int(x) for x in h.tolist() if x >= 0] for h in heads]}
    if ranks_seen != world or not out["rank_sum_ok"]:
        raise RuntimeError(f"collective self-check failed: all-reduce of 1 over {world} ranks gave {ranks_seen}, of the rank "
                           f"ids {rank_sum} (expected {world * (world - 1) // 2})")
    if on_gpu and backend != "gloo" and shared_physical_devices(identities):
        raise RuntimeError(f"collective self-check failed: ranks share a physical device (host, pci domain, bus, device, index) "
                           f"{identities} (one process per GPU)")
    # the step's payload, standalone: `reps` all-reduces, each between two synchronises (what the exchange costs when
    # nothing overlaps it; the timed region overlaps it with compute)
    def sync():
        if on_gpu:
            torch.cuda.synchronize(device)
    times = []
    for i in range(reps + 2):
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(payload)
        sync()
        if i >= 2:   # (the first calls build the communicator's rings / buffers)
            times.append(1e3 * (time.perf_counter() - t0))
    t = torch.tensor([statistics.median(times)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out.update(allreduce_ms_p50=float(t.item()), allreduce_ms_min_rank0=min(times), allreduce_ms_max_rank0=max(times),
               allreduce_reps=reps,
               allreduce_busbw_GBps=(2.0 * (world - 1) / world) * out["payload_bytes"] / (float(t.item()) * 1e-3) / 1e9 if world > 1 else None)
    payload.zero_()   # (the sums of whatever it held are meaningless)
    return out
