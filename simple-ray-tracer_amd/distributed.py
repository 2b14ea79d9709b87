"""One frame over the GPUs of a node: one process per GPU, the BVH replicated, the image cut into tiles.

Ranks draw tile indices from a counter in shared memory (TileCounter: one atomic fetch-add per tile, so a GPU
that lands on cheap tiles simply takes more of them) and render each tile into a slot of their own tile buffer.
There is no exchange between ranks while rendering.  When the counter runs out, the finished tiles are
gathered to rank 0: tile ids travel in one small all_gather, pixels in one point-to-point transfer per peer (on
MI355X every peer has its own direct xGMI link to rank 0, so a flat gather is link-optimal; `backend="nccl"` is
RCCL there).  The same code runs over gloo with CPU tensors, which is how it is tested without GPUs.

The reference does this with OpenMP `schedule(guided)` over image rows inside one process
(reference src/render.cpp:184-187).
"""
import torch
import torch.distributed as dist


def tile_rects(width, height, tile_w, tile_h):
    """Tiles of the frame in row-major order (y up, as the render loops count); edge tiles may be smaller."""
    return [(x, y, min(x + tile_w, width), min(y + tile_h, height))
            for y in range(0, height, tile_h) for x in range(0, width, tile_w)]


class FrameTiler:
    """Tile bookkeeping of one rank.  `device` is where tile buffers and the frame live."""

    def __init__(self, width, height, tile_w, tile_h, device, rank=0, world=1, counter=None, max_in_flight=2):
        self.W, self.H, self.tw, self.th = width, height, tile_w, tile_h
        self.rects = tile_rects(width, height, tile_w, tile_h)
        self.n_tiles = len(self.rects)
        self.device = torch.device(device)
        self.rank, self.world, self.counter = rank, world, counter
        self.max_in_flight = max_in_flight
        if world > 1 and counter is None:
            raise ValueError("more than one rank needs a shared tile counter")
        self.tiles = torch.zeros((self.n_tiles, tile_h, tile_w, 3), dtype=torch.float32, device=self.device)
        self.ids = torch.full((self.n_tiles,), -1, dtype=torch.int32)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device=self.device) if rank == 0 else None
        self.step = 0

    def render(self, render_tile, fence=None):
        """Draws tiles until this frame's counter runs out.  render_tile(rect, slot_tensor) renders (or
        enqueues) one tile into slot_tensor, laid out top row first; fence() returns an object with
        .synchronize() marking the work enqueued so far (None when render_tile is synchronous).  At most
        max_in_flight tiles are outstanding per rank, so a rank cannot hoard tiles it has not started."""
        k = 0
        pending = []
        while True:
            t = self.counter.next(self.step) if self.counter is not None else k
            if t >= self.n_tiles:
                break
            if fence is not None and len(pending) >= self.max_in_flight:
                pending.pop(0).synchronize()
            render_tile(self.rects[t], self.tiles[k])
            if fence is not None:
                pending.append(fence())
            self.ids[k] = t
            k += 1
        return k

    def _place(self, ids, tiles):
        H, th = self.H, self.th
        for j, t in enumerate(ids):
            x0, y0, x1, y1 = self.rects[t]
            self.frame[H - y1:H - y0, x0:x1] = tiles[j, th - (y1 - y0):, :x1 - x0]

    def gather(self, k):
        """Finished tiles of every rank into rank 0's frame (Screen layout: top row first)."""
        if self.world == 1:
            if self.n_tiles == 1 and (self.tw, self.th) == (self.W, self.H):
                self.frame.copy_(self.tiles[0])
            else:
                self._place(self.ids[:k].tolist(), self.tiles)
            self.step += 1
            return
        meta = torch.full((self.n_tiles + 1,), -1, dtype=torch.int32, device=self.device)
        meta[0] = k
        if k:
            meta[1:k + 1] = self.ids[:k].to(self.device)
        metas = [torch.empty_like(meta) for _ in range(self.world)]
        dist.all_gather(metas, meta)
        if self.rank == 0:
            # arm the counter two frames ahead: no rank can be past frame step + 1 before the next gather
            self.counter.reset(self.step + 2)
            host = torch.stack(metas).cpu()
            bufs, reqs = {}, []
            for r in range(1, self.world):
                kr = int(host[r, 0])
                if kr:
                    bufs[r] = torch.empty((kr, self.th, self.tw, 3), dtype=torch.float32, device=self.device)
                    reqs.append(dist.irecv(bufs[r], src=r))
            self._place(self.ids[:k].tolist(), self.tiles)
            for q in reqs:
                q.wait()
            for r, b in bufs.items():
                self._place(host[r, 1:1 + len(b)].tolist(), b)
            owners = host[:, 1:][host[:, 1:] >= 0]
            if sorted(owners.tolist()) != list(range(self.n_tiles)):
                raise RuntimeError("tile distribution lost or duplicated a tile")
        elif k:
            dist.isend(self.tiles[:k].contiguous(), dst=0).wait()
        self.step += 1
