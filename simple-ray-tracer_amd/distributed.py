"""One frame over the GPUs of a node: one process per GPU, the BVH replicated, the image cut into tiles.

Ranks draw tile indices from a counter in shared memory (TileCounter: one atomic fetch-add per tile, so a GPU
that lands on cheap tiles simply takes more of them) and render each tile into a slot of their own tile buffer.
There is no exchange between ranks while rendering.  When the counter runs out, the finished tiles are
gathered to rank 0: who rendered which tile (and what it cost) is posted on a board in shared memory (FrameBoard: the
ranks are processes of one node, so this takes no GPU work, no collective and no device-to-host copy), pixels travel in
one point-to-point transfer per peer, every rank's batched the same way (on MI355X every peer has its own direct xGMI
link to rank 0, so a flat gather is link-optimal; `backend="nccl"` is RCCL there).  The same code runs over gloo with
CPU tensors, which is how it is tested without GPUs.

Hand-out order.  Tiles differ in cost by two orders of magnitude (background against fully covered, every sample
casting 101 shadow rays), and with 8 GPUs the dearest tile of a 4K frame alone is most of a rank's fair share,
so the order decides the frame time.  The counter hands out positions of an ORDER list: centre of the frame
outwards for the first frame, and from then on dearest first by the time each tile took in the previous frame
(every rank times its tiles; the times travel with the tile ids, so all ranks derive the same list).  Measured
per-tile times of the 4K dragon frame put greedy stealing at 0.73 of the ideal 8-GPU frame time in row-major
order, 0.89 centre-out and 0.99 dearest first (tools/sim_tiles.py).

The reference does this with OpenMP `schedule(guided)` over image rows inside one process
(reference src/render.cpp:184-187).
"""
import math
import os
import sys
import time

import torch
import torch.distributed as dist


def tile_rects(width, height, tile_w, tile_h):
    """Tiles of the frame in row-major order (y up, as the render loops count); edge tiles may be smaller."""
    return [(x, y, min(x + tile_w, width), min(y + tile_h, height))
            for y in range(0, height, tile_h) for x in range(0, width, tile_w)]


class FrameTiler:
    """Tile bookkeeping of one rank.  `device` is where tile buffers and the frame live."""

    def __init__(self, width, height, tile_w, tile_h, device, rank=0, world=1, counter=None, max_in_flight=2, board=None):
        self.W, self.H, self.tw, self.th = width, height, tile_w, tile_h
        self.rects = tile_rects(width, height, tile_w, tile_h)
        self.n_tiles = len(self.rects)
        self.device = torch.device(device)
        self.rank, self.world, self.counter = rank, world, counter
        self.max_in_flight = max_in_flight
        if world > 1 and (counter is None or board is None):
            raise ValueError("more than one rank needs a shared tile counter and a frame board")
        self.board = board  # with a board the gather takes the many-rank path even alone (how one GPU rehearses it)
        self.gather_ms = {}  # parts of the last gather() call, host wall clock
        self.tiles = torch.zeros((self.n_tiles, tile_h, tile_w, 3), dtype=torch.float32, device=self.device)
        self.ids = torch.full((self.n_tiles,), -1, dtype=torch.int32)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device=self.device) if rank == 0 else None
        self.step = 0
        cx, cy = width / 2.0, height / 2.0
        self.order = sorted(range(self.n_tiles), key=lambda i: (math.hypot((self.rects[i][0] + self.rects[i][2]) / 2.0 - cx,
                                                                           (self.rects[i][1] + self.rects[i][3]) / 2.0 - cy), i))
        self.cost_us = torch.zeros((self.n_tiles,), dtype=torch.int32)  # of this rank's tiles, by slot
        self._marks = []
        self.current = 0  # ordinal of the tile being rendered: callbacks may use it to alternate between streams

    def render(self, render_tile, fence=None):
        """Draws tiles until this frame's counter runs out.  render_tile(rect, slot_tensor) renders (or
        enqueues) one tile into slot_tensor, laid out top row first; fence() returns an event-like object
        (.synchronize(), .elapsed_time(later)) recorded after the work enqueued so far for the tile self.current
        (callbacks that alternate between streams record on that tile's stream), None when render_tile is
        synchronous.  At most max_in_flight tiles are outstanding per rank, so a rank cannot hoard tiles it has
        not started."""
        k = 0
        pending = []
        self._marks = []
        while True:
            pos = self.counter.next(self.step) if self.counter is not None else k
            if pos >= self.n_tiles:
                break
            t = self.order[pos]
            if fence is not None and len(pending) >= self.max_in_flight:
                pending.pop(0).synchronize()
            self.current = k
            begin = fence() if fence is not None else time.perf_counter()
            render_tile(self.rects[t], self.tiles[k])
            end = fence() if fence is not None else time.perf_counter()
            if fence is not None:
                pending.append(end)
            self._marks.append((begin, end))
            self.ids[k] = t
            k += 1
        return k

    def _measure(self, k):
        """Time of each of this rank's k tiles in microseconds (device time between its two fences)."""
        for j, (begin, end) in enumerate(self._marks[:k]):
            if isinstance(begin, float):
                us = (end - begin) * 1e6
            else:
                end.synchronize()
                us = begin.elapsed_time(end) * 1e3
            self.cost_us[j] = max(1, min(int(us), 2 ** 31 - 1))

    def _reorder(self, ids, costs):
        """Next frame's hand-out order: dearest tile first (ties and unseen tiles by index)."""
        cost = [0] * self.n_tiles
        for t, c in zip(ids, costs):
            cost[t] = c
        self.order = sorted(range(self.n_tiles), key=lambda i: (-cost[i], i))

    def _place(self, ids, tiles):
        H, W, th, tw = self.H, self.W, self.th, self.tw
        if not ids:
            return
        if H % th == 0 and W % tw == 0:
            # whole tiles only: one indexed copy for all of a rank's tiles instead of one small copy per tile
            nx = W // tw
            t = torch.as_tensor(ids, dtype=torch.int64, device=self.frame.device)
            col = t % nx
            row = (H // th - 1) - t // nx  # tile rows count from the bottom of the frame, frame rows from the top
            self.frame.view(H // th, th, nx, tw, 3)[row, :, col] = tiles[:len(ids)]
            return
        for j, t in enumerate(ids):
            x0, y0, x1, y1 = self.rects[t]
            self.frame[H - y1:H - y0, x0:x1] = tiles[j, th - (y1 - y0):, :x1 - x0]

    def _settle(self):
        """Waits until the copies and transfers that read this rank's tile buffer have run: they are ordered on the
        current stream, but the next frame's tiles may be rendered on other streams, which would otherwise be free
        to overwrite a slot that is still being read."""
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def gather(self, k):
        """Finished tiles of every rank into rank 0's frame (Screen layout: top row first)."""
        t0 = time.perf_counter()
        self._measure(k)  # also waits for this rank's tiles, whatever streams they were rendered on
        t1 = time.perf_counter()
        if self.board is None:
            if self.n_tiles == 1 and (self.tw, self.th) == (self.W, self.H):
                self.frame.copy_(self.tiles[0])  # one tile: rendered on the current stream, which orders the copy
            else:
                self._place(self.ids[:k].tolist(), self.tiles)
            t2 = time.perf_counter()
            if self.n_tiles > 1:
                self._settle()
            self.gather_ms = {"wait_tiles": 1e3 * (t1 - t0), "ids": 0.0, "pixels": 1e3 * (t2 - t1), "settle": 1e3 * (time.perf_counter() - t2)}
            self.step += 1
            return
        n = self.n_tiles
        self.board.post(self.step, self.rank, self.ids[:k].numpy(), self.cost_us[:k].numpy())
        records = self.board.collect(self.step)
        t2 = time.perf_counter()
        counts = [len(ids) for ids, _ in records]
        owners = [int(t) for ids, _ in records for t in ids]
        costs = [int(c) for _, cost in records for c in cost]
        if sorted(owners) != list(range(n)):
            raise RuntimeError("tile distribution lost or duplicated a tile")
        self._reorder(owners, costs)
        if self.rank == 0:
            # arm the counter two frames ahead: no rank can be past frame step + 1 before the next gather
            if self.counter is not None:
                self.counter.reset(self.step + 2)
            bufs = {r: torch.empty((counts[r], self.th, self.tw, 3), dtype=torch.float32, device=self.device)
                    for r in range(1, self.world) if counts[r]}
            # one batch: the receives from all peers run side by side, each over its own link
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, r) for r, b in bufs.items()]) if bufs else []
            self._place(self.ids[:k].tolist(), self.tiles)
            for q in reqs:
                q.wait()
            for r, b in bufs.items():
                self._place(records[r][0].tolist(), b)
        elif k:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, self.tiles[:k].contiguous(), 0)]):
                q.wait()
        t3 = time.perf_counter()
        self._settle()
        t4 = time.perf_counter()
        self.gather_ms = {"wait_tiles": 1e3 * (t1 - t0), "ids": 1e3 * (t2 - t1), "pixels": 1e3 * (t3 - t2), "settle": 1e3 * (t4 - t3)}
        if os.environ.get("SRT_BENCH_DEBUG"):
            print(f"[rank {self.rank}] gather: " + ", ".join(f"{k2} {v:.2f} ms" for k2, v in self.gather_ms.items()), file=sys.stderr, flush=True)
        self.step += 1
