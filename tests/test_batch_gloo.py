"""The N>1 path on CPU: two gloo ranks shard a batch of independent images exactly as
bench.py / BASELINE config 5 do on GPUs (image k -> rank k mod world, no data-path
collective), each encoding its shard through the host driver linked against the CPU
emulation of the kernels; the gathered records must equal a single-process run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tests", "emu"))
import torch.distributed as dist
import build_emu, images
from guetzli_amd import batch
from guetzli_amd.encoder import HostLibrary
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
host = HostLibrary(build_emu.HOST_LIB)
get = lambda k: images.shifted(images.crop(32, 32, 300, 150), k)
recs = batch.encode_batch(get, 3, lambda rgb: host.process(rgb, quality=84), rank, world, dist if world > 1 else None)
dt = batch.max_over_ranks(float(rank + 1), dist if world > 1 else None)
assert dt == float(world), dt
# BASELINE config 5 work split (bench.py --config5): 2 images per rank (one at a time here: the
# CPU emulation of the kernels is single-threaded; images in flight are a GPU test)
recs5, secs5 = batch.run_config5(get, 2, lambda rgb: host.process(rgb, quality=84), rank, world,
                                 dist if world > 1 else None, workers=1)
assert len(recs5) == 2 * world and [r["index"] for r in recs5] == list(range(2 * world))
assert all(r["rank"] == r["index"] % world for r in recs5) and secs5 > 0
if rank == 0:
    print("RECORDS", [(r["index"], r["bytes"], r["sha256"], r["rank"]) for r in recs])
    print("CONFIG5", [(r["index"], r["sha256"]) for r in recs5])
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def _run(world, tmp_path):
    script = tmp_path / f"worker{world}.py"
    script.write_text(WORKER.format(root=ROOT))
    import socket
    with socket.socket() as so:   # a free port: the suite runs under pytest-xdist
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    if world == 1:
        env.update(RANK="0", WORLD_SIZE="1")
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                             timeout=600)
    else:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                              f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                              "--master-port", port, str(script)], env=env,
                             capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("RECORDS")][0]
    line5 = [l for l in out.stdout.splitlines() if l.startswith("CONFIG5")][0]
    return eval(line[len("RECORDS "):]), eval(line5[len("CONFIG5 "):])


def test_two_ranks_shard_a_batch(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build_host()
    single, single5 = _run(1, tmp_path)
    double, double5 = _run(2, tmp_path)
    # config 5: 2 images per rank -> world 1 encodes images 0, 1; world 2 images 0..3; the
    # common ones are byte-identical whichever rank (and however many at a time) made them
    assert double5[:2] == single5 and len({h for _, h in double5}) == 4
    assert [r[:3] for r in single] == [r[:3] for r in double]
    assert [r[3] for r in double] == [0, 1, 0]      # image k -> rank k mod 2
    assert len({r[2] for r in double}) == 3         # three different images


FAIL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tests", "emu"))
import torch.distributed as dist
import build_emu, images
from guetzli_amd import batch
from guetzli_amd.encoder import HostLibrary
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
host = HostLibrary(build_emu.HOST_LIB)
get = lambda k: images.shifted(images.crop(32, 32, 300, 150), k)
def proc(rgb):
    if proc.calls == 0 and rank == 1:      # rank 1's first image (index 1) fails inside the driver:
        proc.calls += 1
        return host.process(rgb, quality=70)   # quality < 84 is refused (processor.cc:800-806)
    proc.calls += 1
    return host.process(rgb, quality=84)
proc.calls = 0
try:
    batch.run_config5(get, 2, proc, rank, world, dist, workers=1, fence=dist.barrier)
    sys.stdout.write("RANK %d NO ERROR\n" % rank)
except batch.BatchError as e:
    sys.stdout.write("RANK %d FAILURES %s\n" % (rank, [(f["index"], f["rank"]) for f in e.failures]))   # (one write: two ranks share the pipe)
    sys.stdout.flush()
    dist.barrier(); dist.destroy_process_group()
    sys.exit(3)
"""


def test_a_failed_image_is_reported_by_every_rank_after_the_gather(tmp_path):
    """one image fails on rank 1 (the driver refuses it).  Rank 1 still encodes
    its other image, both ranks pass the fence and the all-gather, BOTH raise BatchError naming image
    1 on rank 1, and the job exits non-zero -- nobody hangs in a collective, nobody exits silently."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build_host()
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29535", str(script)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode != 0, out.stdout + out.stderr
    text = out.stdout + out.stderr
    assert "RANK 0 FAILURES [(1, 1)]" in text, text[-3000:]
    assert "RANK 1 FAILURES [(1, 1)]" in text, text[-3000:]
    assert "NO ERROR" not in text


def test_prepare_runs_ahead_of_the_encodes_and_its_failures_are_the_images():
    """encode_shard_concurrent(prepare=...): the host-only stage in front of an encode (PNG decoding in bench.py's
    from-PNG leg) runs ahead on its own threads, bounded; records keep the input order; an image whose prepare
    fails is reported like one whose encode fails, and the others complete."""
    import threading
    import time
    import pytest
    from guetzli_amd.batch import encode_shard_concurrent, run_config5, BatchError
    lock = threading.Lock()
    state = {"prepared": 0, "consumed": 0, "max_waiting": 0}

    def prepare(k):
        if k == 5:
            raise ValueError("bad PNG")
        with lock:
            state["prepared"] += 1
            state["max_waiting"] = max(state["max_waiting"], state["prepared"] - state["consumed"])
        return ("pixels", k)

    def process(im):
        assert im[0] == "pixels"
        with lock:
            state["consumed"] += 1
        time.sleep(0.01)
        return (b"jpeg%d" % im[1], {})
    recs = encode_shard_concurrent(lambda k: k, range(20), process, workers=2, rank=3, prepare=prepare)
    assert [r["index"] for r in recs] == list(range(20)) and all(r["rank"] == 3 for r in recs)
    assert [r["index"] for r in recs if "error" in r] == [5] and "bad PNG" in recs[5]["error"]
    assert all(r["bytes"] == len(b"jpeg%d" % r["index"]) for r in recs if "error" not in r)
    assert state["prepared"] == 19 and state["max_waiting"] <= 3 * 2 + 2   # (+ the two being handed over)
    with pytest.raises(BatchError) as e:
        run_config5(lambda k: k, 8, process, workers=2, prepare=prepare)
    assert [f["index"] for f in e.value.failures] == [5]
    recs, secs = run_config5(lambda k: k + 10, 4, process, workers=2, prepare=prepare)
    assert [r["index"] for r in recs] == [0, 1, 2, 3] and secs > 0
