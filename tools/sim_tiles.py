#!/usr/bin/env python3
"""Offline: frame time on N GPUs from per-tile costs (gpurun_out/tile_costs.json, tools/tile_costs.py) under greedy
tile stealing, for tile sizes and hand-out orders."""
import heapq
import json
import math

costs = json.load(open("gpurun_out/tile_costs.json"))
W, H = 3840, 2160


def order(kind, tw, th, c):
    nx, ny = -(-W // tw), -(-H // th)
    idx = list(range(len(c)))
    if kind == "row":
        return idx
    if kind == "center":
        return sorted(idx, key=lambda i: math.hypot(((i % nx) + 0.5) * tw - W / 2, ((i // nx) + 0.5) * th - H / 2))
    if kind == "lpt":
        return sorted(idx, key=lambda i: -c[i])
    raise ValueError(kind)


def makespan(c, seq, n):
    free = [0.0] * n
    heapq.heapify(free)
    end = 0.0
    for i in seq:
        t = heapq.heappop(free) + c[i]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


for key, c in costs.items():
    tw, th = map(int, key.split("x"))
    for kind in ("row", "center", "lpt"):
        seq = order(kind, tw, th, c)
        print(f"{key:8s} {kind:7s} total {sum(c):6.1f}  " + "  ".join(f"N={n}: {makespan(c, seq, n):6.1f} ms (eff {sum(costs['480x270']) / n / makespan(c, seq, n):.2f})" for n in (2, 4, 8)))
