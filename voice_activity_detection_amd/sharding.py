"""Partitioning of one audio stream over ranks (one process per GPU).

Every output frame depends on at most 3 earlier audio frames (MFCC alignment),
the +-2 neighbouring MFCC rows of its own slice (delta) and the 29 following
feature rows (window), and the reference already treats slices as independent
calls (core/process_data.py:214-232).  Ranks therefore take contiguous,
slice-aligned frame ranges, regenerate or re-read their own 3-frame lead-in and
never exchange data; the per-window results are gathered on the host.
"""
from .common import FRAMES, SLICE_MAX

# Frames a rank needs beyond its own shard to score every window that STARTS in the shard: 29 more
# feature rows for the window, and 2 more MFCC rows so that the deltas of those 29 rows see their real
# neighbours rather than the replicated edge of a truncated slice.
HALO_FRAMES = (FRAMES - 1) + 2


def shard_range(total_frames, world, rank, slice_len=SLICE_MAX):
    """(first_frame, n_frames) of ``rank``'s contiguous shard; boundaries fall on slice starts."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    if slice_len <= 0:
        # the whole stream is ONE slice (as mfcc_delta_stream treats it): cutting it anywhere would restart
        # the pre-emphasis and the delta edges at the cut, so it cannot be sharded
        if world > 1:
            raise ValueError('a stream processed as one slice (slice_len <= 0) cannot be sharded over %d ranks' % world)
        slice_len = max(total_frames, 1)
    n_slices = (total_frames + slice_len - 1) // slice_len
    base, rem = divmod(n_slices, world)
    s0 = rank * base + min(rank, rem)
    s1 = s0 + base + (1 if rank < rem else 0)
    first = min(s0 * slice_len, total_frames)
    last = min(s1 * slice_len, total_frames)
    return first, last - first


def shard_plan(total_frames, world, rank, slice_len=SLICE_MAX, frames=FRAMES):
    """Everything one rank needs: dict with
    ``first``        first frame of the shard (a slice start),
    ``n_own``        frames of the shard,
    ``n_load``       frames to run the front end over (shard + halo, clipped to the stream),
    ``first_window`` / ``n_windows``  the windows that start in the shard.
    Feature rows ``[0, n_windows + frames - 1)`` of the loaded range are exact; the last rows of
    the halo may carry a truncated slice's delta edge and are never read by those windows."""
    first, n_own = shard_range(total_frames, world, rank, slice_len)
    n_load = min(n_own + HALO_FRAMES, total_frames - first) if n_own else 0
    first_window, n_windows = window_owner_ranges(total_frames, world, slice_len, frames)[rank]
    return {'first': first, 'n_own': n_own, 'n_load': n_load, 'first_window': first_window, 'n_windows': n_windows}


def window_owner_ranges(total_frames, world, slice_len=SLICE_MAX, frames=FRAMES):
    """[(first_window, n_windows)] per rank when every rank scores the windows that START in its
    shard; a rank needs ``frames - 1`` feature rows of halo from its successor for the last ones."""
    out = []
    n_win = max(total_frames - frames + 1, 0)
    for r in range(world):
        first, n = shard_range(total_frames, world, r, slice_len)
        lo = min(first, n_win)
        hi = min(first + n, n_win)
        out.append((lo, hi - lo))
    return out


def gather_posteriors(p_local, world, dist=None):
    """Host-side gather of per-window posteriors (5 bytes/frame of result traffic; no device collective)."""
    import torch
    host = p_local.detach().cpu()
    if dist is None or world == 1:
        return host
    parts = [None] * world
    dist.all_gather_object(parts, host)
    return torch.cat(parts, 0)


def all_gather_posteriors(p_local, world, dist=None):
    """Collective gather that leaves the result where the inputs are (device-resident with the NCCL = RCCL
    backend over xGMI, host tensors with gloo): every rank ends up with the posteriors of all windows.
    Shards differ in length by less than two slices, so the ranks exchange their lengths, pad to the
    longest and trim after one ``all_gather_into_tensor`` (9 bytes per window; 10 MB per 100 h and rank).
    With a process group the collectives run for any world size, a single rank included."""
    import torch
    if dist is None:
        return p_local
    n = torch.tensor([p_local.shape[0]], dtype=torch.int64, device=p_local.device)
    counts = torch.empty(world, dtype=torch.int64, device=p_local.device)
    dist.all_gather_into_tensor(counts, n)
    counts = [int(c) for c in counts.cpu()]
    longest = max(counts)
    padded = torch.zeros((longest,) + tuple(p_local.shape[1:]), dtype=p_local.dtype, device=p_local.device)
    padded[:p_local.shape[0]] = p_local
    out = torch.empty((world * longest,) + tuple(p_local.shape[1:]), dtype=p_local.dtype, device=p_local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * longest: r * longest + counts[r]] for r in range(world)], 0)
