"""Data-parallel sharding of a crop batch across the GPUs of one node (no counterpart in the reference,
whose inference is single-GPU batch-1).  One process per GPU, weights replicated, crops dealt round-robin, and ONE
collective per search step: an all-gather of the fixed-size per-crop result records (`vstar_result`, 193,568 B) so that
every rank can take the next-step decision deterministically.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU
is used by the world_size-2 tests."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin deal: item i goes to rank i % world."""
    return list(range(rank, n_items, world))


def pad_count(n_items: int, world: int) -> int:
    """Records per rank after padding so that every rank contributes the same count to the all-gather."""
    return (n_items + world - 1) // world


def allgather_records(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """local: [pad_count, R] records of this rank's shard (rows beyond the shard are padding).  Returns [n_items, R]
    in the ORIGINAL item order on every rank."""
    if not dist.is_initialized():
        return local[:n_items]
    world = dist.get_world_size(group)          # a one-rank group still runs the collective (RCCL self-check on one GPU)
    per = local.shape[0]
    gathered = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    gathered = gathered.view(world, per, *local.shape[1:])
    # item i lives at [i % world, i // world]
    idx = torch.arange(n_items, device=local.device)
    return gathered[idx % world, idx // world]


def engine_comm_init(engine, group=None) -> None:
    """Gives `engine` (VstarEngine) its own RCCL communicator over the ranks of the torch process group: rank 0 draws the
    128-byte unique id (vstar_comm_unique_id) and broadcasts it through the group's store — any backend, the id is host bytes —
    then every rank enters vstar_comm_init.  Afterwards VSM._score_sharded gathers the records with vstar_allgather_results on
    the engine's stream (no torch.distributed on the data path)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], world, rank)


def maybe_engine_comm(vsm, group=None) -> bool:
    """The DEFAULT data-path collective of the crop-sharded entry points on GPUs (round 6; `--engine-comm off` keeps
    torch.distributed): with a multi-rank nccl group and an engine that has the comm entry points, the engine gets its own RCCL
    communicator (engine_comm_init) and PROVES it before use — one record per rank, filled with a rank-specific pattern, goes through
    vstar_allgather_results and through torch.distributed's all_gather_into_tensor, and the two results must be bit-identical on
    every rank.  Any failure — missing entry points, communicator set-up, a mismatching self-check — leaves the torch.distributed
    gather in place, on EVERY rank: the outcome is agreed on with an all-reduce (min), so no rank can take the other collective.
    Hardware status: one-rank RCCL self-check on every GPU run (tests/test_rccl_selfcheck_gpu.py); the >= 2-rank path is exercised by
    tests/test_multi_gpu_nccl.py, which needs a box with two GPUs (no such box has run it yet: no scaling curve exists)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"):
        return False
    eng = getattr(vsm, "engine", None)
    world = dist.get_world_size(group)
    ok = 1
    try:
        if eng is None or not hasattr(eng, "comm_init"):
            raise RuntimeError("engine has no comm entry points")
        if getattr(eng, "comm_world", 0) != world:     # (an engine keeps its communicator: set up once)
            engine_comm_init(eng, group)
        # self-check against the torch collective on a pattern no rank shares
        rank = dist.get_rank(group)
        dev = f"cuda:{eng.device}"
        local = (torch.arange(eng.lib_result_floats(), dtype=torch.float32, device=dev) * (rank + 1) + rank).reshape(1, -1).contiguous()
        mine = eng.allgather_results(local)
        ref = torch.empty((world, local.shape[1]), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(ref, local, group=group)
        if not torch.equal(mine, ref):
            raise RuntimeError("vstar_allgather_results disagrees with torch.distributed.all_gather_into_tensor")
    except Exception as exc:            # noqa: BLE001 — fall back, but on EVERY rank
        import warnings
        warnings.warn(f"engine communicator not available ({type(exc).__name__}: {exc}); using torch.distributed")
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{torch.cuda.current_device()}")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    vsm.use_engine_comm = bool(int(flag.item()))
    return vsm.use_engine_comm


# ---------------- what to deal over the ranks: crops of a step, or whole samples ----------------
# Measured step times of ONE rank at 1 / 2 / 4 / 8 / 16 / 32 crops per call (ms; BENCH_r05 `per_rank_shape` + the headline step): the
# engine is far from linear at small batches (18.6 ms for one crop, 7.2 ms per crop at 32), which is what decides the question.
DEFAULT_STEP_MS = {1: 18.6, 2: 25.1, 4: 39.0, 8: 68.8, 16: 124.0, 32: 232.0}
COLLECTIVE_MS = 0.25        # one all-gather of <= 32 records of 194 KB over xGMI + the rank rendezvous (estimate; not yet measured on > 1 GPU)


def step_ms(n_crops: int, table=None) -> float:
    """Step time of one rank scoring n_crops crops in one call: the measured table, linear between its points, linear beyond."""
    t = sorted((table or DEFAULT_STEP_MS).items())
    if n_crops <= 0:
        return 0.0
    if n_crops <= t[0][0]:
        return t[0][1]
    for (n0, m0), (n1, m1) in zip(t, t[1:]):
        if n_crops <= n1:
            return m0 + (m1 - m0) * (n_crops - n0) / (n1 - n0)
    return t[-1][1] * n_crops / t[-1][0]


def choose_shard(n_samples: int, world: int, window: int, table=None, crops_per_search_step: float = 4.0) -> str:
    """`--shard auto`: "crops" or "samples" for a run of n_samples searches on `world` ranks with `window` searches in flight per rank
    group (visual_search.py --window).  crops_per_search_step: what one search contributes to an engine step — 1 for the reference's
    schedule, ~4 with the stream driver's speculation of a node's likely next crops (vstar_amd/search.py; the default of the entry
    points).

    crops:   every rank walks all samples; a step scores min(window * world, remaining) crops, dealt round-robin, + one collective.
    samples: searches are dealt round-robin; a rank steps through its own n_samples / world searches, `window` at a time, no collective.
    Both schedules give a rank ~window crops per step while work lasts; they differ at the edges: with fewer searches than ranks x
    window, sample sharding leaves ranks idle or under-filled (a single search cannot use a second GPU at all), crop sharding keeps
    every rank at the same batch but pays the collective and rank-0-paced host decisions on every step.  The model below prices both
    with the measured step-time table and picks the cheaper; ties go to "samples" (no collective on the data path)."""
    if world <= 1 or n_samples <= 0:
        return "samples"
    window = max(1, window)
    steps_per_search = 6.0                                   # mean visited patches of a depth-3 search (tests/golden/search_paths.json)
    # crops: windows of window * world searches
    k = max(1.0, float(crops_per_search_step))
    in_flight = min(n_samples, window * world)
    rounds = -(-n_samples // (window * world))
    per_rank = int(-(-(in_flight * k) // world))
    t_crops = rounds * steps_per_search * (step_ms(per_rank, table) + COLLECTIVE_MS)
    # samples: the slowest rank has ceil(n_samples / world) searches, window at a time
    mine = -(-n_samples // world)
    t_samples = -(-mine // window) * steps_per_search * step_ms(int(-(-(min(mine, window) * k) // 1)), table)
    return "crops" if t_crops < t_samples else "samples"


def reorder_gathered(gathered: torch.Tensor, world: int, n_items: int) -> torch.Tensor:
    """[world * per, R] rank-major (what an all-gather of round-robin shards returns) -> [n_items, R] in item order."""
    per = gathered.shape[0] // world
    g = gathered.view(world, per, *gathered.shape[1:])
    idx = torch.arange(n_items, device=gathered.device)
    return g[idx % world, idx // world]


def allgather_numpy(local: np.ndarray, n_items: int, device: str = "cpu") -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    return allgather_records(t, n_items).cpu().numpy()


# ---------------- process-group lifecycle of the entry points (visual_search.py, vstar_bench_eval.py, bench.py) ----------------
def init_from_env(backend: str | None = None):
    """Joins the process group described by the torchrun / torch.distributed.run environment (RANK, WORLD_SIZE, LOCAL_RANK,
    MASTER_ADDR, MASTER_PORT).  Returns (world, rank, local_rank); a plain `python script.py` launch (no WORLD_SIZE, or 1) is
    world 1 and creates no group.  Backend: "nccl" (= RCCL over xGMI) when a GPU is visible, else "gloo"."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: VSTAR_DIST_BACKEND forces the backend, VSTAR_DIST_DEVICE puts every rank on ONE device — several ranks of the REAL
    # engine on a single GPU over gloo (RCCL refuses two ranks on one device): tests/test_two_ranks_one_gpu.py
    backend = os.environ.get("VSTAR_DIST_BACKEND") or backend
    if os.environ.get("VSTAR_DIST_DEVICE") is not None:
        local_rank = int(os.environ["VSTAR_DIST_DEVICE"])
    if (world > 1 or os.environ.get("VSTAR_FORCE_PROCESS_GROUP") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def finalize(ok: bool = True) -> None:
    """Leaves the process group.  The barrier belongs to the SUCCESS path only: a rank that is unwinding an exception (for example
    the reference-semantics IndexError of a search whose decode emitted no [LOC]) must not wait for peers that are still inside a
    collective — it tears its group down so that the job fails instead of hanging until the RCCL timeout."""
    if dist.is_available() and dist.is_initialized():
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def gather_objects(obj, world: int):
    """Every rank's `obj`, in rank order, on every rank (sample-level data parallelism: per-rank metric lists)."""
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
