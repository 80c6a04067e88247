"""bench.py's own control flow at world size 2 on CPU: `bench.py --emulate` runs main() -- the
process group, the timed bracket (barrier + max over ranks), the rank-0-only legs, the
config-5 leg with its all-gather, the final barrier -- over gloo with the test-suite's CPU
emulation of the kernels on tiny images, launched exactly as the driver launches the GPU run
(python -m torch.distributed.run --nproc-per-node 2).  It checks the JSON contract of the one
line rank 0 prints; it measures nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _free_port():
    """A port nobody listens on right now (the suite runs under pytest-xdist: a fixed port made two of these
    tests collide whenever they landed on different workers at the same time)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def _run(world, extra=()):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build_host()   # (built once here, not by two ranks at the same time)
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    args = [os.path.join(ROOT, "bench.py"), "--emulate", "--gpus", str(world), "--steps", "1",
            "--warmup", "0", "--no-cpu-baseline"] + list(extra)
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", port] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_main_two_ranks_gloo():
    r = _run(2)
    for k in CONTRACT:
        assert k in r, k
    assert r["n_gpus"] == 2 and r["steps"] == 1 and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["higher_is_better"] is True
    assert r["value_1080p"] > 0 and r["ms_per_step_1080p"] > 0 and r["config_1080p"]["steps"] == 1
    assert "configs[2]" in r["config"]["workload"] and "configs[1]" in r["config_1080p"]["workload"]
    for key in ("roofline", "roofline_1080p"):
        assert r[key]["bound"] == "hbm" and r[key]["peak"] == 8000.0
        assert abs(r[key]["frac"] - r[key]["achieved"] / r[key]["peak"]) < 1e-3
    c5 = r["other_configs"]["config5_slice"]
    assert c5["images"] == 2 * c5["images_per_gpu"] and c5["distinct_outputs"] == c5["images"]
    assert r["scale_value"] == c5["value"]
    assert c5["n_ranks_seen"] == 2 and c5["ranks_in_records"] == [0, 1]
    assert c5["images_per_rank"] == {"0": c5["images_per_gpu"], "1": c5["images_per_gpu"]}
    # every rank's host-CPU binding is on the line (all-gathered): disjoint shares of the cores of this container
    bind = c5["binding_per_rank"]
    assert [b["rank"] for b in bind] == [0, 1] and all(b["n_host_cpus"] >= 1 and b["how"] for b in bind)
    from guetzli_amd.affinity import parse_cpulist
    assert not set(parse_cpulist(bind[0]["host_cpus"])) & set(parse_cpulist(bind[1]["host_cpus"]))
    assert r["value_1mpix"] > 0 and r["config_1mpix"]["iterations"] >= 1
    assert "cpu_baseline" not in r and "batch_one_gpu" not in r   # N = 1 only


def test_bench_main_eight_ranks_gloo():
    """The run the driver launches first on an 8-GPU node (`--gpus 8` under torch.distributed.run
    with 8 processes), as a dry run on this container's cores: 8 gloo ranks, one emulated image
    each, the config-5 leg with its all-gather over 8 ranks."""
    r = _run(8, ["--images-per-gpu", "1", "--batch-images", "0"])
    for k in CONTRACT:
        assert k in r, k
    assert r["n_gpus"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["value_4k"] == r["value"] and r["value_1080p"] > 0 and r["value_workload"]
    c5 = r["other_configs"]["config5_slice"]
    assert c5["images"] == 8 and c5["n_ranks_seen"] == 8 and c5["ranks_in_records"] == list(range(8))
    assert c5["images_per_rank"] == {str(k): 1 for k in range(8)} and c5["distinct_outputs"] == 8
    assert r["scale_value"] == c5["value"]


def test_bench_config5_only_two_ranks_gloo():
    r = _run(2, ["--config5", "--images-per-gpu", "1"])
    assert r["n_gpus"] == 2 and r["config"]["images"] == 2 and r["value"] == r["config"]["value"]


def test_worker_pool_is_sized_from_the_affinity_mask():
    """A rank bound to two cores gets a two-thread pool, whatever the machine has (VERDICT r3:
    std::thread::hardware_concurrency ignores the mask Env.bind_cpus sets)."""
    from guetzli_amd import build as gzbuild
    lib = gzbuild.build_host()
    code = f"import ctypes; print(ctypes.CDLL({lib!r}).gzh_worker_pool_size())"
    env = {k: v for k, v in os.environ.items() if k != "GZ_HOST_THREADS"}
    two = subprocess.run(["taskset", "-c", "0-1", sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=120)
    assert two.returncode == 0, two.stderr
    assert int(two.stdout) == 2
    one = subprocess.run(["taskset", "-c", "0", sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=120)
    assert int(one.stdout) == 1


def test_bench_two_ranks_on_two_cores_finishes():
    """The multi-GPU launch on a box with ONE host core per rank ("one core per GPU
    on this box"): `taskset -c 0-1` around the world-2 dry run of config 5 -- every rank binds to
    its single core, sizes its pool from it, and the run completes."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build_host()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    cmd = ["taskset", "-c", "0-1", sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--emulate", "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--config5", "--images-per-gpu", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["config"]["n_ranks_seen"] == 2 and r["config"]["host_cores_per_rank"] == 1
