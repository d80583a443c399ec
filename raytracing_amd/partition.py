"""Samples-per-pixel partitioning across GPUs.

Frames are independent given (LoopNum, randOrigin) and the image is their mean, so the
frames of a render are dealt out to the ranks in contiguous blocks, every rank keeps a plain
fp32 sum of its own samples, and ONE sum-reduce to rank 0 finishes the image.  The reference
has no counterpart (it is single-GPU, and its 8-bit running mean cannot be split)."""
import numpy as np

from . import rand_origins


def frame_block(rank, world, step, frames_per_step):
    """LoopNum values (1-based) rendered by `rank` in step `step`: a contiguous block."""
    first = (step * world + rank) * frames_per_step + 1
    return first, frames_per_step


def rand_origin_stream(seed, total_frames):
    """randOrigin of LoopNum 1..total_frames; every rank derives the same stream from the seed,
    so the set of samples does not depend on the number of ranks."""
    return rand_origins(seed, total_frames)


def reduce_sum_to_root(sum_tensor, dist=None):
    """Sum-reduce the per-rank W*H*3 fp32 sums to rank 0 (RCCL over xGMI on GPUs, gloo on CPU).
    A group of one rank still goes through the backend (the collective is then a copy in place)."""
    if dist is not None and dist.is_initialized():
        dist.reduce(sum_tensor, dst=0, op=dist.ReduceOp.SUM)
    return sum_tensor


def mean_from_sum(total_sum, total_frames):
    return total_sum / np.float32(total_frames)
