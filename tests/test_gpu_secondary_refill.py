"""Two choices the render loop makes by the size of a rectangle, each forced the other way here and the parity tests that
send on reflection, transmission and environment rays run once more in a child process.
The reflection-ray kernel exists twice: with a fixed stride over the path queue (small rectangles) and fed like the shadow
kernel, in chunks with lane refill (rectangles of 2^24 samples and more): SRT_SECONDARY_REFILL.
A frame with reflections runs its launches on one stream, or - short frames of deep trees - as two chains on two streams,
the light stages beside the next depth's trace: SRT_OVERLAP."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_parity_with_the_refill_kernel_forced():
    env = dict(os.environ, SRT_SECONDARY_REFILL="1")
    r = subprocess.run([sys.executable, "-u", "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_random_scenes.py"),
                        "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "150", "--timeout-method", "thread", "-k",
                        "recursive or reflection or transparency or environment or render_matches or random_scenes or final_color or tiles"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["0", "1"])
def test_parity_with_the_two_stream_frame_forced(overlap):
    env = dict(os.environ, SRT_OVERLAP=overlap)
    r = subprocess.run([sys.executable, "-u", "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_random_scenes.py"),
                        "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "150", "--timeout-method", "thread", "-k",
                        "recursive or reflection or environment or render_matches or random_scenes or final_color or tiles or multisample or dragon_standin_frame"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
