"""Data-parallel encode over the GPUs of one node: one process per GPU, images sharded on the
batch axis, ONE all-gather of visual tokens at the projector boundary (RCCL over xGMI when the
process group's backend is "nccl"; the same code runs over gloo on CPU in the tests).

The reference has no collective on this path: images are independent through the
whole encoder and projector (eval-mode BatchNorm `mci.py:901-907`, per-pixel LayerNorm
`mci.py:617-623`, per-image attention `mci.py:661-685`; a list of images is a loop of B=1 calls,
`mobileclip_encoder.py:78-83`), so sharding needs no communication inside the path.  The gather
exists only because the consumer (the LLM prefill, `llava_arch.py:146-332`) wants every image's
tokens.

xGMI is point-to-point (7 links per GPU), so the gather is issued as a single
`all_gather_into_tensor` of each rank's whole shard (12.6 MB for 8 images x 256 x 3072 bf16) -
one large message per peer, never per-image messages.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n items: the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_images(images: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(images.shape[0], rank, world)
    return images[lo:hi]


def all_gather_into(out: torch.Tensor, local: torch.Tensor, group=None) -> torch.Tensor:
    """`dist.all_gather_into_tensor(out, local)` for whatever backend the group runs on.  RCCL ("nccl") and gloo-on-CPU take the tensors
    as they are; gloo has no all-gather for DEVICE tensors, so there the message is staged through the host (round 5: the only way to run
    the real tower under world size > 1 on a ONE-GPU box - RCCL refuses two ranks on one device - and a debugging aid elsewhere; never the
    production path, where the backend is nccl and the tokens go GPU to GPU over xGMI)."""
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host_out = torch.empty(out.shape, dtype=out.dtype, device="cpu")
        dist.all_gather_into_tensor(host_out, local.contiguous().cpu(), group=group)
        out.copy_(host_out)
        return out
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_tokens(local: torch.Tensor, global_batch: int, group=None, force: bool = False) -> torch.Tensor:
    """local [b_r, T, H] on each rank (b_r from `shard_bounds`) -> [global_batch, T, H] on every rank,
    in image order.  Equal shards take the single-message path; ragged shards are padded to the
    largest shard for the collective and trimmed afterwards.  force: issue the collective at world size 1 too (tests / `--force-dist`)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    bmax = max(counts)
    tail = tuple(local.shape[1:])
    if min(counts) == bmax:
        out = torch.empty((global_batch,) + tail, dtype=local.dtype, device=local.device)
        return all_gather_into(out, local, group=group)
    padded = torch.zeros((bmax,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * bmax,) + tail, dtype=local.dtype, device=local.device)
    all_gather_into(buf, padded, group=group)
    return torch.cat([buf[r * bmax: r * bmax + counts[r]] for r in range(world)], 0)


def encode_images_data_parallel(encode_fn: Callable[[torch.Tensor], torch.Tensor], images_local: torch.Tensor,
                                global_batch: int, group=None, gather: bool = True) -> torch.Tensor:
    """`encode_fn` is this rank's encode_images (tower -> projector) on its own shard.  With
    `gather=False` the rank-local tokens are returned (throughput benchmarking of the shardable
    part); with `gather=True` every rank gets all `global_batch` images' tokens."""
    local = encode_fn(images_local)
    if not gather:
        return local
    return all_gather_tokens(local, global_batch, group=group)


TOWER_WIDTH = 3072       # FastViTHD token width (cls_ratio 2.0 x 1536, mci.py:1403) = the projector's input width


def gather_side(llm_hidden: int) -> str:
    """Which side of the projector the all-gather sits on: the collective moves b x T x width bytes per rank, so
    it goes where the tokens are NARROWER - after the projector for FastVLM-0.5B / 1.5B (H = 896 / 1536 < 3072), before it for
    FastVLM-7B (H = 3584 > 3072: 12.6 MB instead of 14.7 MB per rank at 8 images, and every rank then projects all
    global_batch images - 12 GFLOP per image that hide behind nothing but are 0.3 % of the encode)."""
    return "before" if llm_hidden > TOWER_WIDTH else "after"


def encode_images_sharded(tower_fn: Callable[[torch.Tensor], torch.Tensor], projector_fn: Callable[[torch.Tensor], torch.Tensor],
                          images_local: torch.Tensor, global_batch: int, llm_hidden: int, group=None,
                          side: Optional[str] = None) -> torch.Tensor:
    """encode_images over sharded images with the gather on the cheaper side of the projector (`gather_side`):
    "after":  gather(projector(tower(local)))       - what `encode_images_data_parallel` does with a fused encode_fn;
    "before": projector(gather(tower(local)))       - the projector runs on all `global_batch` images on every rank."""
    side = side or gather_side(llm_hidden)
    if side not in ("before", "after"):
        raise ValueError(f"side must be 'before' or 'after', got {side!r}")
    tokens = tower_fn(images_local)
    if side == "after":
        return all_gather_tokens(projector_fn(tokens), global_batch, group=group)
    return projector_fn(all_gather_tokens(tokens, global_batch, group=group))


def encode_images_tower_sharded(vision_tower, mm_projector, images_local: torch.Tensor, global_batch: int, group=None,
                                side: Optional[str] = None) -> torch.Tensor:
    """`encode_images_sharded` for a tower + projector pair: both legs run in the library whenever `builder.library_projector` holds -
    side "after": ONE fused `fvhd_encode_images` call per rank, then the gather; side "before" (H > 3072, FastVLM-7B): `fvhd_encode`,
    the gather of the 3072-wide tokens, then `fvhd_project` on all `global_batch` images (never torch.nn.Linear)."""
    from . import builder
    hidden = mm_projector[0].out_features if isinstance(mm_projector, torch.nn.Sequential) else getattr(mm_projector, "out_features", TOWER_WIDTH)
    side = side or gather_side(hidden)
    if side == "after":
        return all_gather_tokens(builder.encode_images(vision_tower, mm_projector, images_local), global_batch, group=group)
    if side != "before":
        raise ValueError(f"side must be 'before' or 'after', got {side!r}")
    return builder.project(vision_tower, mm_projector, all_gather_tokens(vision_tower(images_local), global_batch, group=group))

