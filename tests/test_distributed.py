"""The multi-GPU frame path (tile stealing through the shared counter, gather of finished tiles to rank 0) on
CPU: two gloo ranks, tiles rendered by the oracle."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

srt = importlib.import_module("simple-ray-tracer_amd")
tiling = importlib.import_module("simple-ray-tracer_amd.distributed")


def test_tile_rects_cover_frame_once():
    W, H = 100, 72
    rects = tiling.tile_rects(W, H, 32, 24)
    cover = np.zeros((H, W), np.int32)
    for x0, y0, x1, y1 in rects:
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all() and len(rects) == 4 * 3


def test_tile_counter_hands_out_each_index_once():
    name = f"/srt_test_counter_{os.getpid()}"
    a = srt.TileCounter(name, True)
    b = srt.TileCounter(name, False)
    got = [a.next(0), b.next(0), a.next(0), b.next(1), b.next(0)]
    assert got == [0, 1, 2, 0, 3]
    a.reset(0)
    assert b.next(0) == 0 and a.next(64) == 1  # slots wrap at 64 frames
    b.close()
    a.close()


def test_single_rank_tiler_assembles_frame():
    import torch
    ft = tiling.FrameTiler(10, 6, 4, 4, "cpu")

    def render_tile(rect, slot):
        x0, y0, x1, y1 = rect
        slot.zero_()
        for y in range(y0, y1):
            for x in range(x0, x1):
                slot[4 - 1 - (y - y0), x - x0, :] = float(y * 10 + x)

    ft.gather(ft.render(render_tile))
    want = np.array([[y * 10 + x for x in range(10)] for y in reversed(range(6))], np.float32)
    assert np.array_equal(ft.frame[..., 0].numpy(), want)


def _run_ranks(tmp_path, ranks, *worker_args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "frame.npz"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(out), *worker_args]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def test_three_ranks_one_without_tiles(tmp_path):
    """Uneven split: rank 2 only asks for tiles after ranks 0 and 1 have drawn them all, so it takes part in the
    gather with nothing to send, frame after frame."""
    z = _run_ranks(tmp_path, 3, "2", "2")
    assert np.array_equal(z["frame"].view(np.uint32), z["whole"].view(np.uint32))
    counts = z["counts"]  # [rank, frame]
    # (how ranks 0 and 1 share the tiles between them depends on the machine's load: either may draw them all)
    assert counts.shape == (3, 2) and (counts[2] == 0).all()
    assert (counts.sum(0) == int(z["n_tiles"])).all()
    assert (z["orders"] == z["orders"][0]).all()


def test_two_ranks_gloo(tmp_path):
    z = _run_ranks(tmp_path, 2, "3")
    assert np.array_equal(z["frame"].view(np.uint32), z["whole"].view(np.uint32))  # assembled frame == one-piece frame
    counts = z["counts"]  # [rank, frame]
    assert (counts.sum(0) == int(z["n_tiles"])).all()  # every frame: all tiles, none twice
    assert z["whole"].max() > 0.05
    orders = z["orders"]  # next frame's hand-out order as each rank derived it from the gathered tile times
    assert (orders == orders[0]).all() and sorted(orders[0].tolist()) == list(range(int(z["n_tiles"])))


def test_hand_out_order():
    ft = tiling.FrameTiler(100, 72, 32, 24, "cpu")  # 4 x 3 tiles
    assert sorted(ft.order) == list(range(12))
    assert ft.order[0] == 5 and ft.order[-1] in (3, 11)  # the centre tile first, a far corner last
    ft._reorder([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [5, 1, 1, 9, 1, 1, 7, 1, 1, 1, 1, 1])
    assert ft.order[:3] == [3, 6, 0] and ft.order[3:] == [1, 2, 4, 5, 7, 8, 9, 10, 11]


@pytest.mark.parametrize("shape", [(96, 72, 32, 24), (100, 72, 32, 24), (64, 64, 64, 64), (64, 48, 16, 48)])
def test_tiles_land_where_their_rectangles_say(shape):
    """_place puts a rank's tiles (any order) into the frame: whole-tile frames take one indexed copy, frames with
    ragged edge tiles one copy per tile; both must equal the per-rectangle assignment (tile rows count from the
    bottom of the frame, frame rows from the top)."""
    import torch
    W, H, tw, th = shape
    ft = tiling.FrameTiler(W, H, tw, th, "cpu")
    ids = list(np.random.default_rng(3).permutation(ft.n_tiles))
    tiles = torch.rand(ft.n_tiles, th, tw, 3)
    want = torch.zeros(H, W, 3)
    for j, t in enumerate(ids):
        x0, y0, x1, y1 = ft.rects[t]
        want[H - y1:H - y0, x0:x1] = tiles[j, th - (y1 - y0):, :x1 - x0]
    ft._place([int(t) for t in ids], tiles)
    assert torch.equal(ft.frame, want)


def test_frame_board_posts_and_collects():
    """FrameBoard: every rank posts the tiles it rendered, everybody reads everybody's; slots are reused every four frames
    without any reset; a missing post times out instead of hanging."""
    name = f"/srt_test_board_{os.getpid()}"
    a = srt.FrameBoard(name, True, 2, 12)
    b = srt.FrameBoard(name, False, 2, 12)
    with pytest.raises(srt.SrtError, match="another world size"):
        srt.FrameBoard(name, False, 3, 12)
    for frame in range(11):
        mine = np.arange(frame % 5, dtype=np.int32) + frame
        a.post(frame, 0, mine, mine * 10)
        b.post(frame, 1, [11], [frame])
        for board in (a, b):
            rec = board.collect(frame, timeout_s=5)
            assert np.array_equal(rec[0][0], mine) and np.array_equal(rec[0][1], mine * 10)
            assert rec[1][0].tolist() == [11] and rec[1][1].tolist() == [frame]
    a.post(11, 0, [], [])
    with pytest.raises(srt.SrtError, match="timed out"):
        a.collect(11, timeout_s=0.05)  # rank 1 never posted frame 11
    with pytest.raises(srt.SrtError, match="bad post"):
        a.post(12, 5, [], [])
    b.close()
    a.close()
