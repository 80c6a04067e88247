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
if rank == 0:
    print("RECORDS", [(r["index"], r["bytes"], r["sha256"], r["rank"]) for r in recs])
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def _run(world, tmp_path):
    script = tmp_path / f"worker{world}.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    if world == 1:
        env.update(RANK="0", WORLD_SIZE="1")
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                             timeout=600)
    else:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                              f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                              "--master-port", "29533", str(script)], env=env,
                             capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("RECORDS")][0]
    return eval(line[len("RECORDS "):])


def test_two_ranks_shard_a_batch(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build_host()
    single = _run(1, tmp_path)
    double = _run(2, tmp_path)
    assert [r[:3] for r in single] == [r[:3] for r in double]
    assert [r[3] for r in double] == [0, 1, 0]      # image k -> rank k mod 2
    assert len({r[2] for r in double}) == 3         # three different images
