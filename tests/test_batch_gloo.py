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
