"""Row-strip image partition across the GPUs of one node and the final gather.

The reference treats image rows as independent work items (rayon `par_chunks_mut` over rows,
all-is-cubes-render/src/raytracer/renderer.rs:537-555). Here the rows are grouped into strips
of `strip_rows`; strip s is rendered by rank (s % world_size) (interleaved, because rows differ
wildly in cost: sky rows vs. geometry rows). The scene is replicated on every GPU; the only
exchange step is one gather of the finished RGBA8 strips to rank 0 (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" for the CPU tests) followed by a de-interleave.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

# rows per strip: the kernel's work tile (8 x 8 pixels). 1080 rows are 135 such strips -- 17 or 16 per rank at N = 8 (136 / 128 rows) -- where 16-row strips (rounds 1-5)
# made 67.5 of them: 9 or 8 per rank, 144 / 128 rows, and the frame waits for its slowest rank (a rank's share at N = 8: 0.0593 -> 0.0567 ms, profiles/r06_experiments.txt T)
STRIP_ROWS = 8


def partition_rows(height: int, strip_rows: int, n_parts: int, part: int) -> List[int]:
    """Global row numbers rendered by `part`, in increasing order (== aic_partition_rows)."""
    return [y for y in range(height) if (y // strip_rows) % n_parts == part]


def max_partition_rows(height: int, strip_rows: int, n_parts: int) -> int:
    return max(len(partition_rows(height, strip_rows, n_parts, p)) for p in range(n_parts))


def gather_strips(local: torch.Tensor, height: int, width: int, strip_rows: int, group=None) -> Optional[torch.Tensor]:
    """`local`: this rank's compacted rows as uint8 [rows_local, width, 4] (device or CPU).
    Returns on rank 0 the gathered buffer [world, max_rows, width, 4]; None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_rows = max_partition_rows(height, strip_rows, world)
    padded = local
    if local.shape[0] != max_rows:
        padded = torch.zeros((max_rows, width, 4), dtype=torch.uint8, device=local.device)
        padded[: local.shape[0]] = local
    if world == 1:
        return padded.unsqueeze(0)
    if dist.get_backend(group) == "nccl":
        # one collective: every rank's strips land contiguously in rank 0's buffer
        out = torch.empty((world, max_rows, width, 4), dtype=torch.uint8, device=local.device) if rank == 0 else None
        dist.gather(padded, gather_list=list(out.unbind(0)) if rank == 0 else None, dst=0, group=group)
        return out
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gather_list=gathered, dst=0, group=group)
    return torch.stack(gathered, 0) if rank == 0 else None


def assemble_strips_torch(gathered: torch.Tensor, height: int, width: int, strip_rows: int) -> torch.Tensor:
    """Reference de-interleave with tensor indexing (used on CPU tensors and to cross-check the
    device kernel `aic_assemble_strips`)."""
    world, max_rows = gathered.shape[0], gathered.shape[1]
    y = torch.arange(height, device=gathered.device)
    strip = y // strip_rows
    part = strip % world
    lrow = (strip // world) * strip_rows + (y % strip_rows)
    return gathered[part, lrow]


class StripGatherPipeline:
    """The per-frame exchange step, `depth` frames deep: while frame i's strips travel to rank 0,
    frame i+1 is already being traced. A ring of `depth` slots, each a local strip buffer
    [max_rows, width, 4] (every rank) and a gather target [world, max_rows, width, 4] (rank 0):

        buf = pipe.local[slot]            # render this rank's strips into it
        pipe.submit(slot)                 # asynchronous gather of the slot to rank 0
        ...
        g = pipe.retire(slot)             # before the slot is reused: wait for its gather;
                                          # rank 0 gets the gathered strips back (de-interleave them)

    The renderer works on HIP streams of its own, which torch's stream semantics do not cover. `retire` therefore
    records an event behind the finished gather and hands it to `wait_event` (the renderer's `aic_wait_event`:
    everything it queues afterwards -- the de-interleave of this slot, the next trace into its strip buffer -- waits
    for the event on the device; the host does not block). Without a `wait_event` hook the host waits for that one
    event. Buffers are allocated once.

    `frames` > 1 makes a slot hold that many consecutive frames, gathered by ONE collective (`frame_buffer(slot, k)` is
    frame k's strip buffer; `retire` then returns [world, frames, max_rows, width, 4] and `assemble` de-interleaves frame
    k of it). At 8 ranks a rank's share of a 1080p frame is traced in about 0.1 ms, which is also what one gather call
    costs the host: several frames per collective keep the exchange step off the frame period."""

    def __init__(self, height: int, width: int, strip_rows: int, device, depth: int = 2, group=None, wait_event=None, frames: int = 1):
        self.group = group
        self.wait_event = wait_event
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.height, self.width, self.strip_rows, self.depth = height, width, strip_rows, depth
        self.max_rows = max_partition_rows(height, strip_rows, self.world)
        self.frames = max(1, int(frames))
        self.buffers = [torch.zeros((self.frames, self.max_rows, width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]
        full = [torch.empty((self.world, self.frames, self.max_rows, width, 4), dtype=torch.uint8, device=device) if self.rank == 0 else None
                for _ in range(depth)]
        self.gather_lists = [list(g.unbind(0)) if g is not None else None for g in full]  # built once: submit() is on the frame path
        # one frame per slot: the slot IS the frame's strip buffer, and what comes back is [world, max_rows, width, 4]
        self.local = [b[0] for b in self.buffers] if self.frames == 1 else self.buffers
        self.gathered = [(g[:, 0] if self.frames == 1 else g) if g is not None else None for g in full]
        self._index = None
        self.work = [None] * depth
        self.order: List[int] = []  # slots with a gather in flight, oldest first

    def submit(self, slot: int) -> None:
        assert self.work[slot] is None, "slot still in flight: retire it first"
        self.work[slot] = dist.gather(self.buffers[slot], gather_list=self.gather_lists[slot], dst=0, group=self.group, async_op=True)
        self.order.append(slot)

    def retire(self, slot: int) -> Optional[torch.Tensor]:
        w = self.work[slot]
        if w is None:
            return None
        w.wait()  # NCCL: torch's current stream now waits for the collective (not the host)
        if self.local[slot].is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.local[slot].device))
            if self.wait_event is not None:
                self.wait_event(ev.cuda_event)
                self._events = getattr(self, "_events", [])[-8:] + [ev]  # keep the handles alive until they have fired
            else:
                ev.synchronize()
        self.work[slot] = None
        self.order.remove(slot)
        return self.gathered[slot]

    def frame_buffer(self, slot: int, k: int = 0) -> torch.Tensor:
        """Frame k of a slot: this rank's strips [max_rows, width, 4], contiguous."""
        return self.buffers[slot][k]

    def assemble(self, gathered: torch.Tensor, k: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """De-interleave frame k of what `retire` returned (rank 0) into a [height, width, 4] frame with tensor indexing
        (for one frame per slot the renderer's own `aic_assemble_strips` does this on its stream)."""
        g = gathered[:, k] if gathered.dim() == 5 else gathered
        if self._index is None or self._index[0].device != g.device:
            y = torch.arange(self.height, device=g.device)
            strip = y // self.strip_rows
            self._index = (strip % self.world, (strip // self.world) * self.strip_rows + (y % self.strip_rows))
        if out is None:
            return g[self._index[0], self._index[1]]
        return out.copy_(g[self._index[0], self._index[1]])

    def oldest(self) -> Optional[int]:
        return self.order[0] if self.order else None
