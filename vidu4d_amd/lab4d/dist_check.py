"""Self-validation of the first N > 1 launch (frame-parallel Stage-3).

The reference trusts its launcher (/root/reference/lab4d/train.py:28-36 sets the device from LOCAL_RANK and calls
init_process_group; lab4d/utils/gpu_utils.py:6-128 hands GPUs to workers) and finds out about a bad rendezvous from a hang.
No multi-GPU box is reachable from the build container, so the first real 8-GPU run of this code is the driver's: it carries
its own evidence.  `collective_self_check` runs right after init_process_group and returns a small dict that goes on the
bench's JSON line (key "rccl") and into train.py's first log line:

  world / backend / ranks_seen (all-reduce of 1) / rank_sum_ok (all-reduce of the rank ids) / device_of_each_rank /
  payload_bytes / allreduce_ms_p50, _min, _max over `reps` standalone all-reduces of the step's payload, each bracketed by
  a device synchronise / frames_of_each_rank_head (all-gather of the first frame ids every rank renders).

It aborts (RuntimeError on every rank) when a rank's HIP device is not its LOCAL_RANK, when two ranks share a device under
RCCL, or when a sum comes back wrong -- before any timing is taken.
"""
from __future__ import annotations

import statistics
import time

import torch


def collective_self_check(dist, device, local_rank: int, payload: torch.Tensor, frames_head, backend: str = "nccl",
                          reps: int = 20) -> dict:
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = device.type == "cuda"
    if on_gpu and backend != "gloo":
        cur = torch.cuda.current_device()
        if cur != local_rank or device.index != local_rank:
            raise RuntimeError(f"rank {rank}: HIP device {cur} (tensor device {device}) is not LOCAL_RANK {local_rank} -- "
                               "one process per GPU, device = LOCAL_RANK (launch with torch.distributed.run --nproc-per-node N)")
    one = torch.ones(1, device=device, dtype=torch.float64)
    dist.all_reduce(one)
    ids = torch.tensor([float(rank)], device=device, dtype=torch.float64)
    dist.all_reduce(ids)
    devs = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(devs, torch.tensor([device.index if on_gpu else -1], dtype=torch.int64, device=device))
    head = list(frames_head)[:4] + [-1] * max(0, 4 - len(list(frames_head)[:4]))
    heads = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(heads, torch.tensor(head, dtype=torch.int64, device=device))
    ranks_seen, rank_sum = int(round(one.item())), int(round(ids.item()))
    devices = [int(d.item()) for d in devs]
    out = {"world": world, "backend": "rccl" if backend == "nccl" else backend, "ranks_seen": ranks_seen,
           "rank_sum_ok": rank_sum == world * (world - 1) // 2, "device_of_each_rank": devices,
           "payload_bytes": int(payload.numel() * payload.element_size()),
           "frames_of_each_rank_head": [[int(x) for x in h.tolist() if x >= 0] for h in heads]}
    if ranks_seen != world or not out["rank_sum_ok"]:
        raise RuntimeError(f"collective self-check failed: all-reduce of 1 over {world} ranks gave {ranks_seen}, of the rank "
                           f"ids {rank_sum} (expected {world * (world - 1) // 2})")
    if on_gpu and backend != "gloo" and len(set(devices)) != world:
        raise RuntimeError(f"collective self-check failed: ranks share a device {devices} (one process per GPU)")
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
