"""Batch sharding of independent images over the GPUs of one node (BASELINE
config 5): image k -> rank k mod world, one process per GPU, no data-path collective.  The
process group (RCCL on GPUs, gloo in the CPU tests) is used only for the work split's
bookkeeping: a barrier, the max-over-ranks of the elapsed time and an all-gather of one small
result record per image -- what the reference's golden test does with `xargs -P`
(tests/golden_test.sh:24-26)."""
import hashlib
import time


def shard(n_images, rank, world):
    """Indices of the images rank `rank` encodes."""
    return list(range(rank, n_images, world))


def max_over_ranks(seconds, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def encode_batch(get_image, n_images, process, rank=0, world=1, dist=None):
    """Encodes this rank's shard with `process(rgb) -> (jpeg_bytes, info)` and returns the
    records of ALL images (index, bytes, sha256, seconds, rank), ordered by index, on every
    rank."""
    mine = []
    for k in shard(n_images, rank, world):
        t0 = time.perf_counter()
        jpg, _ = process(get_image(k))
        mine.append({"index": k, "bytes": len(jpg), "sha256": hashlib.sha256(jpg).hexdigest(),
                     "seconds": time.perf_counter() - t0, "rank": rank})
    if dist is None or world == 1:
        return mine
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    return sorted((r for part in gathered for r in part), key=lambda r: r["index"])


def hint_in_flight(n):
    """Tells the device library how many images this process is about to run at once (gz_hint_images_in_flight:
    no priority streams among a batch's streams).  Quietly nothing when the product library is not the one in use
    (the CPU suite's emulation)."""
    try:
        import guetzli_amd
        guetzli_amd.load().lib.gz_hint_images_in_flight(int(n))
    except Exception:
        pass


def encode_concurrent(images, process, workers=4):
    """Several independent images on ONE GPU at the same time: one host thread per image in
    flight (the C++ driver releases the GIL for the whole encode; every image has its own
    device context and stream).  A single encode keeps the GPU busy about a third of the time
    -- the rest is the serial search logic on the host -- so images in flight overlap one
    image's host work with another's kernels.  Returns [(jpeg_bytes, info)] in input order."""
    from concurrent.futures import ThreadPoolExecutor
    hint_in_flight(workers)
    try:
        with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
            return list(ex.map(process, images))
    finally:
        hint_in_flight(1)


class BatchError(RuntimeError):
    """One or more images of a batch failed; .failures = [{"index", "rank", "error"}, ...] as every
    rank saw them after the gather."""

    def __init__(self, failures):
        self.failures = failures
        super().__init__("batch: " + "; ".join(
            f"image {f['index']} on rank {f['rank']}: {f['error']}" for f in failures))


def encode_shard_concurrent(get_image, indices, process, workers, rank=0, prepare=None):
    """This rank's images, `workers` of them in flight on its GPU; records as encode_batch's.  An
    image that fails does not take the others with it: its record is {"index", "rank", "error"} --
    the caller gathers the records of all ranks first and raises afterwards (run_config5), so a
    multi-GPU job says WHICH image failed WHERE instead of dying at a barrier.
    prepare (optional): host-only work in front of an image's encode -- decoding its PNG bytes -- done AHEAD
    on threads of its own, at most 3 x workers images beyond the ones being encoded: an image thread then
    finds its pixels waiting instead of decoding them while its slot on the GPU stands empty (the first
    `workers` decodes are the only ones nothing hides).  A failure of prepare is the image's failure."""
    indices = list(indices)
    ahead = None
    if prepare is not None:
        import threading
        from concurrent.futures import ThreadPoolExecutor as _Pool
        ahead = {"pool": _Pool(max_workers=max(1, workers)), "futs": {}, "next": 0, "lock": threading.Lock()}

        def submit_more(limit):
            with ahead["lock"]:
                while ahead["next"] < len(indices) and len(ahead["futs"]) < limit:
                    k = indices[ahead["next"]]
                    ahead["futs"][k] = ahead["pool"].submit(lambda k=k: prepare(get_image(k)))
                    ahead["next"] += 1
        submit_more(3 * max(1, workers))

    def fetch(k):
        if ahead is None:
            return get_image(k)
        with ahead["lock"]:
            fut = ahead["futs"].get(k)
        if fut is None:   # (not reached while images are taken in order)
            return prepare(get_image(k))
        try:
            return fut.result()
        finally:
            with ahead["lock"]:
                ahead["futs"].pop(k, None)
            submit_more(3 * max(1, workers))

    def one(k):
        t0 = time.perf_counter()
        try:
            jpg, _ = process(fetch(k))
        except Exception as e:   # (the C++ driver's failures arrive as RuntimeError)
            return {"index": k, "rank": rank, "error": f"{type(e).__name__}: {e}",
                    "seconds": time.perf_counter() - t0}
        return {"index": k, "bytes": len(jpg), "sha256": hashlib.sha256(jpg).hexdigest(),
                "seconds": time.perf_counter() - t0, "rank": rank}
    from concurrent.futures import ThreadPoolExecutor
    hint_in_flight(workers)
    try:
        with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
            return list(ex.map(one, indices))
    finally:
        hint_in_flight(1)
        if ahead is not None:
            ahead["pool"].shutdown(wait=True)


def raise_on_failures(records):
    """After the gather: every rank sees the same records, so every rank raises (and exits non-zero)."""
    failures = [{"index": r["index"], "rank": r["rank"], "error": r["error"]} for r in records if "error" in r]
    if failures:
        raise BatchError(failures)


def run_config5(get_image, images_per_gpu, process, rank=0, world=1, dist=None, workers=4,
                fence=None, device=None, prepare=None):
    """BASELINE config 5 ("a batch of independent images sharded 8-per-GPU across the GPUs of
    one node"): image k -> rank k mod world (k < images_per_gpu * world), every rank keeps
    `workers` of its images in flight on its GPU, no data-path collective.  The process group
    serves the work split only: barrier, max-over-ranks of the elapsed time, all-gather of one
    small record per image.  Returns (records of all images ordered by index, seconds).  A failed
    image raises BatchError on EVERY rank, after the gather, naming image, rank and error."""
    n = images_per_gpu * world
    if fence:
        fence()
    t0 = time.perf_counter()
    mine = encode_shard_concurrent(get_image, shard(n, rank, world), process, workers, rank, prepare)
    if fence:
        fence()
    seconds = max_over_ranks(time.perf_counter() - t0, dist, device)
    if dist is None or world == 1:
        raise_on_failures(mine)
        return mine, seconds
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    records = sorted((r for part in gathered for r in part), key=lambda r: r["index"])
    raise_on_failures(records)   # (behind the fence and the gather: no rank is left waiting in a collective)
    return records, seconds
