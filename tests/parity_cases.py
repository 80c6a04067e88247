"""Parity cases shared by the CPU-emulation run (`-m "not gpu"`, tests/test_kernels_emu.py)
and the real-GPU run (`-m gpu`, tests/test_gpu_parity.py): the same kernels, the same
checks, through the same C ABI -- only the library differs."""
import numpy as np

import images
from checkers import assert_bits_equal, oracle

RNG_SEED = 20260921

SIGMAS_BR = [(1.2, 0.0), (7.46953768697, -0.00457628248637), (3.734768843485, -0.271277366628),
             (1.8673844217425, 0.147068973249), (10.6666499623, 0.0),
             (9.24456601467, -0.0724948220913), (2.3770330432, -0.0724948220913),
             (9.04353323561, -0.0724948220913), (1.72547472444, 1.0)]


def case_block_kernels(L, n=1500):
    rng = np.random.default_rng(RNG_SEED)
    blocks = rng.integers(-4096, 4097, size=(n, 64)).astype(np.int16)
    blocks[: n // 2][rng.random((n // 2, 64)) < 0.7] = 0
    blocks = np.concatenate([blocks, np.full((1, 64), 32767, np.int16),
                             np.full((1, 64), -32768, np.int16), np.zeros((2, 64), np.int16)])
    got = L.idct_blocks(blocks)
    exp = np.stack([oracle.idct_block(b) for b in blocks])
    assert_bits_equal(got, exp, "idct blocks")
    px = np.concatenate([rng.integers(-128, 128, size=(n, 64)).astype(np.int16),
                         np.full((1, 64), -128, np.int16), np.full((1, 64), 127, np.int16)])
    got = L.fdct_blocks(px)
    exp = np.stack([oracle.fdct_block(b) for b in px])
    assert_bits_equal(got, exp, "fdct blocks")


def case_dct_double(L, n=1500):
    """dct_double.cc: FP64 block transforms, bit for bit."""
    rng = np.random.default_rng(RNG_SEED + 8)
    blocks = np.concatenate([
        rng.integers(-2048, 2048, size=(n, 64)).astype(np.float64),          # coefficient-like
        rng.random((n // 2, 64)) * 255.0,                                      # pixel-like
        rng.standard_normal((n // 2, 64)) * 10.0 ** rng.integers(-30, 30, (n // 2, 64)),
        np.zeros((1, 64)), -np.zeros((1, 64)), np.full((1, 64), 255.0)])
    for inverse in (False, True):
        got = L.dct_double_blocks(blocks, inverse=inverse)
        exp = np.stack([oracle.dct_double(b, inverse) for b in blocks])
        assert_bits_equal(got, exp, f"dct_double inverse={inverse}")


def case_downsample_component(L, w, h, x0=0, y0=0):
    """ToFloatPixels + SetDownsampledCoefficients (output_image.cc:99-121,265-300), the two
    users of dct_double.cc on the YUV420 path, for the subsampling factors guetzli uses."""
    rgb = images.crop(w, h, x0, y0)
    co = oracle.encode_rgb(rgb)
    for c in range(3):
        px = L.component_to_float_pixels(co[c], w, h)
        assert_bits_equal(px, oracle.to_float_pixels(co[c], w, h), f"ToFloatPixels c={c}")
    for fx, fy in ((2, 2), (1, 1)):
        eu, ev = oracle.downsample_chroma(co, w, h, fx, fy)
        for c, exp in ((1, eu), (2, ev)):
            px = L.component_to_float_pixels(co[c], w, h)
            got = L.component_set_downsampled(px, fx, fy)
            assert_bits_equal(got, exp, f"SetDownsampledCoefficients c={c} {fx}x{fy} {w}x{h}")


def case_encode_quantize_reconstruct(L, w, h, x0=0, y0=0):
    rng = np.random.default_rng(RNG_SEED + w)
    rgb = images.crop(w, h, x0, y0)
    with L.context(rgb, 1.0) as ctx:
        co = ctx.encode_rgb()
        exp = oracle.encode_rgb(rgb)
        assert_bits_equal(co, exp, "encode_rgb")
        q = np.stack([rng.integers(1, 12, size=64), rng.integers(1, 20, size=64),
                      rng.integers(1, 20, size=64)]).astype(np.int32)
        for qq in (None, q):
            cq = ctx.quantize(qq)
            ecq, esrgb, elin = oracle.reconstruct(exp, w, h, qq)
            assert_bits_equal(cq, ecq, "quantize")
            srgb, lin = ctx.reconstruct()
            assert_bits_equal(srgb, esrgb, "reconstruct srgb")
            assert_bits_equal(lin, elin, "reconstruct linear")
        # block updates
        idx = np.unique(np.array([0, ctx.nb - 1, ctx.nb // 2], np.int32))   # must be distinct
        blocks = rng.integers(-300, 300, size=(len(idx), 3, 64)).astype(np.int16)
        ctx.set_coeff_blocks(idx, blocks)
        ecq2 = ecq.copy()
        for i, b in enumerate(idx):
            ecq2[:, b, :] = blocks[i]
        assert_bits_equal(ctx.get_coeffs(), ecq2, "set_coeff_blocks")


def case_blur(L, w, h, configs=SIGMAS_BR):
    rng = np.random.default_rng(RNG_SEED + 7 * w + h)
    plane = (rng.random((h, w)) * 255).astype(np.float32)
    rgb = np.zeros((h, w, 3), np.uint8)
    with L.context(rgb, 1.0) as ctx:
        for s, br in configs:
            got = ctx.probe_blur(plane, s, br)
            exp = oracle.blur(plane, s, br)
            assert_bits_equal(got, exp, f"blur sigma={s} br={br} {w}x{h}")


def _linear_pair(w, h, x0, y0, qscale):
    rgb = images.crop(w, h, x0, y0)
    co = oracle.encode_rgb(rgb)
    q = np.full((3, 64), qscale, np.int32)
    cq, _, lin1 = oracle.reconstruct(co, w, h, q)
    lut = oracle.srgb_table()
    lin0 = lut[rgb].astype(np.float32).transpose(2, 0, 1).copy()
    return rgb, co, cq, lin0, lin1


def case_stages(L, w, h, x0=40, y0=60, qscale=6):
    """opsin -> separate_frequencies -> mask -> diffmap, each against the oracle."""
    rgb, co, cq, lin0, lin1 = _linear_pair(w, h, x0, y0, qscale)
    with L.context(rgb, 1.0) as ctx:
        x0_, x1_ = oracle.opsin(lin0), oracle.opsin(lin1)
        assert_bits_equal(ctx.probe_opsin(lin0), x0_, "opsin(orig)")
        assert_bits_equal(ctx.probe_opsin(lin1), x1_, "opsin(cand)")
        for xyb in (x0_, x1_):
            got = ctx.probe_separate_frequencies(xyb)
            exp = oracle.separate_frequencies(xyb)
            names = ["lf0", "lf1", "lf2", "mf0", "mf1", "mf2", "hf0", "hf1", "uhf0", "uhf1"]
            for i, nm in enumerate(names):
                if nm == "mf2":
                    continue   # dead plane in the reference (wmul[5] == 0), never computed
                assert_bits_equal(got[i], exp[i], f"separate_frequencies {nm}")
        for a, b in ((x0_, x1_), (x0_, x0_)):
            gm, gdc = ctx.probe_mask(a, b)
            em, edc = oracle.mask(a, b)
            assert_bits_equal(gm, em, "mask")
            assert_bits_equal(gdc, edc, "mask_dc")
        gd, gs = ctx.probe_diffmap(lin0, lin1)
        ed, es = oracle.diffmap(lin0, lin1)
        assert_bits_equal(gd, ed, "diffmap")
        assert gs == np.float32(es)


def case_compare(L, w, h, x0=0, y0=0, qscales=(1, 3, 9), target=0.971769):
    """The drop-in call sequence of TryQuantMatrix: encode -> quantize(q) -> compare."""
    rgb = images.crop(w, h, x0, y0) if max(w, h) <= 444 else images.tiled(w, h)
    oc = oracle.comparator(rgb, target)
    with L.context(rgb, target) as ctx:
        co = ctx.encode_rgb()
        assert_bits_equal(co, oracle.encode_rgb(rgb), "encode_rgb")
        for qs in qscales:
            q = np.full((3, 64), qs, np.int32)
            cq = ctx.quantize(q)
            dist, dm, bm = ctx.compare()
            edist, edm = oc.compare(cq)
            assert_bits_equal(dm, edm, f"distmap q={qs}")
            assert dist == edist
            # block maxima + weights
            pad = np.zeros((ctx.bh * 8, ctx.bw * 8), np.float32)
            pad[:h, :w] = edm
            ebm = pad.reshape(ctx.bh, 8, ctx.bw, 8).max(axis=(1, 3)).reshape(-1)
            assert_bits_equal(bm, ebm, "block max")
            for direction in (1, -1):
                for r in (1, 3):
                    assert_bits_equal(ctx.block_weights(direction, r, 1.0),
                                      oc.block_weights(direction, r, 1.0, edm),
                                      "block weights")
    oc.close()


def case_block_search(L, w, h, x0=300, y0=150, qs=3, target=0.971769):
    """Phase A of SelectFrequencyMasking through gz_block_zeroing_orders."""
    rgb = images.crop(w, h, x0, y0) if max(w, h) <= 444 else images.tiled(w, h)
    oc = oracle.comparator(rgb, target)
    with L.context(rgb, target) as ctx:
        co = ctx.encode_rgb()
        cq = ctx.quantize(np.full((3, 64), qs, np.int32))
        off, idx, err = ctx.block_zeroing_orders()
        eoff, eidx, eerr = oc.block_zeroing_orders(cq, co)
        assert_bits_equal(off, eoff, "candidate offsets")
        assert_bits_equal(idx, eidx, "candidate coefficient indices")
        assert_bits_equal(err, eerr, "candidate errors")
        assert off[-1] > 0
    oc.close()


def case_global_order(L, w, h, x0=300, y0=150, qs=3, target=0.971769):
    """Phase B's global candidate order on the device (processor.cc:622-663): the construction
    from the reference's definition, the device-side block weights
    (ComputeBlockErrorAdjustmentWeights, pinned through the oracle) and max_block_error
    bookkeeping, and single-coefficient edits."""
    rng = np.random.default_rng(RNG_SEED + 3 * w + h)
    rgb = images.crop(w, h, x0, y0) if max(w, h) <= 444 else images.tiled(w, h)
    oc = oracle.comparator(rgb, target)
    with L.context(rgb, target) as ctx:
        ctx.encode_rgb()
        cq = ctx.quantize(np.full((3, 64), qs, np.int32))
        off, idx, err = ctx.block_zeroing_orders()
        dist, dm, bmax = ctx.compare()
        nb = ctx.nb
        cnt = np.diff(off)
        max_err = np.zeros(nb, np.float32)
        ctx.order_reset()
        for direction, use_dm in ((1, False), (1, True), (-1, True), (-1, True)):
            next_cand = (rng.integers(0, 1000, nb) % (cnt + 1)).astype(np.int32)
            for radius in (1, 2, 4):
                zero = np.zeros_like(dm)
                wgt = oc.block_weights(direction, radius, 1.0, dm if use_dm else zero)
                # the reference's construction loop
                exp = []
                btc = 0
                for b in range(nb):
                    if wgt[b] == 0:
                        continue
                    e = err[off[b]:off[b + 1]]
                    at = next_cand[b]
                    if direction > 0:
                        vals = (e[at:] - max_err[b]) / wgt[b]
                        btc += at < cnt[b]
                    else:
                        vals = (max_err[b] - e[:at][::-1]) / wgt[b]
                        btc += at > 0
                    exp.extend((b, v) for v in vals.astype(np.float32))
                limit = np.float32(0.75) * np.float32(target)
                total, got_btc, below = ctx.order_build_auto(direction, radius, 1.0, use_dm,
                                                             next_cand, limit=float(limit))
                assert total == len(exp) and got_btc == btc, (total, len(exp), got_btc, btc)
                got = ctx.order_fetch(0, total)
                if total:
                    eb = np.array([b for b, _ in exp], np.int32)
                    ev = np.array([v for _, v in exp], np.float32)
                    assert_bits_equal(got["block"], eb, "order blocks")
                    assert_bits_equal(got["val"], ev, "order vals")
                    assert below == int((ev < limit).sum())
                # the two-halves form (construction enqueued ahead of the wait) = the one-call form
                ctx.order_build_auto_begin(direction, radius, 1.0, use_dm, next_cand, limit=float(limit))
                assert ctx.order_build_auto_end() == (total, btc, below)
                assert_bits_equal(ctx.order_fetch(0, total), got, "order after _begin/_end")
                # explicit-weights entry point gives the same order
                t2, b2, _ = ctx.order_build(direction, next_cand, max_err, wgt)
                assert (t2, b2) == (total, btc)
                assert_bits_equal(ctx.order_fetch(0, t2), got, "gz_order_build vs _auto")
            # leave the last radius' weights in place and advance, as the driver does
            thr = np.float32(rng.random() * 0.3)
            ctx.order_build_auto(direction, 4, 1.0, use_dm, next_cand)
            ctx.order_advance(float(thr), direction)
            wgt = oc.block_weights(direction, 4, 1.0, dm if use_dm else np.zeros_like(dm))
            max_err = (max_err + (wgt * thr) * np.float32(direction)).astype(np.float32)
        # whole-block steps (processor.cc:704-736): zero / restore the next candidates
        def quantize(raw, q):
            r = int(np.fmod(raw, q))
            d = q - r if 2 * r > q else (-q - r if -2 * r > q else -r)
            return np.int16(raw + d)
        co = ctx.get_coeffs()
        orig = ctx.encode_rgb()
        for direction in (1, -1):
            exp = co.copy()
            if direction > 0:
                next_cand = (rng.integers(0, 1000, nb) % (cnt + 1)).astype(np.int32)
                counts = (rng.integers(0, 1000, nb) % (cnt - next_cand + 1)).astype(np.int32)
            else:
                next_cand = (rng.integers(0, 1000, nb) % (cnt + 1)).astype(np.int32)
                counts = (rng.integers(0, 1000, nb) % (next_cand + 1)).astype(np.int32)
            sel = np.flatnonzero(counts > 0).astype(np.int32)
            for b in sel:
                for j in range(counts[b]):
                    p = next_cand[b] + j if direction > 0 else next_cand[b] - 1 - j
                    ix = int(idx[off[b] + p])
                    c, k = ix // 64, ix % 64
                    ob = orig[c, b].astype(np.int64)
                    newval = 0 if direction > 0 else int(quantize(int(ob[k]), qs))
                    precious = False
                    if newval == 0 and k in (1, 8):
                        hf = sum(abs(int(ob[i])) for i in range(3, 64) if not ((i & 7) < 3 and i < 24))
                        precious = abs(int(ob[k])) >= (4 if hf < 60 else 8)
                    if not precious:
                        exp[c, b, k] = newval
            ctx.order_build_auto(direction, 1, 1.0, True, next_cand)   # uploads next_cand
            qs_all = np.full((3, 64), qs, np.int32)
            before = ctx.jpeg_histograms(qs_all)                        # also: the symbols' quantiser
            ctx.apply_candidate_steps(direction, sel, counts[sel])
            delta = ctx.steps_histogram_delta()
            co = ctx.get_coeffs()
            assert_bits_equal(co, exp, f"apply_candidate_steps direction {direction}")
            # the statistics change the steps report == a recount of the whole image
            after = ctx.jpeg_histograms(qs_all)
            assert_bits_equal(delta, after[1].astype(np.int64) - before[1].astype(np.int64),
                              f"steps_histogram_delta direction {direction}")
        cq = co
        # single-coefficient edits == block scatter
        pos = rng.choice(3 * nb * 64, size=min(500, nb), replace=False).astype(np.int32)
        val = rng.integers(-50, 50, pos.size).astype(np.int16)
        ctx.apply_coeff_edits(pos, val)
        expc = cq.copy().reshape(-1)
        expc[pos] = val
        assert_bits_equal(ctx.get_coeffs().reshape(-1), expc, "apply_coeff_edits")
    oc.close()


def case_patched_candidate_planes(L, w, h, x0=0, y0=0, qs=3, target=0.971769, rounds=3, expect_ahead=True):
    """gz_config.patch_reconstruct through the C ABI alone: after bulk steps on a minority of the block positions
    (gz_apply_candidate_steps) and single-coefficient edits elsewhere (gz_apply_coeff_edits), a Compare that relies on
    the patched linear planes (and, in mode 2, checks them against a full reconstruction itself) gives the distance,
    distance map and per-block maxima, bit for bit, of the same coefficients put in place as a whole (gz_set_coeffs:
    full reconstruction).  Ragged sizes patch partial blocks at the right / bottom edge.  A change of more than half
    of the positions, or a whole-image writer in between, drops the patches' claim."""
    rng = np.random.default_rng(RNG_SEED + 5 * w + h)
    rgb = images.crop(w, h, x0, y0) if max(w, h) <= 444 else images.tiled(w, h)
    q = np.full((3, 64), qs, np.int32)
    with L.context(rgb, target) as ctx:
        cfg = ctx.set_config(patch_reconstruct=2)
        expect_ahead = expect_ahead and cfg.opsin_ahead != 0 and cfg.single_stream != 1   # (the suite's forced modes)
        ctx.encode_rgb()
        ctx.quantize(q)
        off, idx, err = ctx.block_zeroing_orders()
        nb = ctx.nb
        cnt = np.diff(off)
        ctx.order_reset()
        ctx.compare()                                    # the full reconstruction: the planes are the candidate's
        next_cand = np.zeros(nb, np.int32)
        for r in range(rounds):
            ctx.order_build_auto(1, 1, 1.0, True, next_cand)     # (uploads next_cand)
            ctx.jpeg_histograms(q)                               # (the steps' statistics path)
            some = rng.random(nb) < (0.3 if r < rounds - 1 else 0.8)
            counts = np.where(some, np.minimum(cnt - next_cand, 1 + rng.integers(0, 3, nb)), 0).astype(np.int32)
            sel = np.flatnonzero(counts > 0).astype(np.int32)
            before = L.compare_counters(all=True)
            ctx.apply_candidate_steps(1, sel, counts[sel])
            ctx.steps_histogram_delta()
            next_cand = next_cand + counts
            pos = rng.choice(3 * nb * 64, size=min(40, nb // 4), replace=False).astype(np.int32)   # (fewer than half of the positions)
            ctx.apply_coeff_edits(pos, rng.integers(-40, 40, pos.size).astype(np.int16))
            got = ctx.compare()
            patched, checked, compares, ahead, ahead_checked = (a - b for a, b in zip(L.compare_counters(all=True), before))
            # (the last round touches most positions: no patches, the Compare reconstructs)
            assert (patched, checked, compares) == ((1, 1, 1) if 2 * sel.size <= nb else (0, 0, 1)), (patched, checked, sel.size, nb)
            # (the opsin image kept ahead as well -- gz_config.opsin_ahead, a context alone on its device -- and checked)
            assert ahead == ahead_checked and ahead in ((0, 1) if patched else (0,))
            if patched and expect_ahead:
                assert ahead == 1
            co = ctx.get_coeffs()
            ctx.set_coeffs(co)
            before = L.compare_counters()
            exp = ctx.compare()
            assert L.compare_counters()[0] == before[0]
            for g, e, what in zip(got, exp, ("distance", "distance map", "block maxima")):
                assert_bits_equal(np.asarray(g, np.float32), np.asarray(e, np.float32), f"patched {what}, round {r}")


ZIGZAG_NATURAL = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
                  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def expected_jpeg_histograms(cq, q):
    """BuildDCHistograms / BuildACHistograms (jpeg_data_writer.cc:241-275) from their
    definition, on quantised values cq / q (C truncation)."""
    cq = cq.astype(np.int64)
    qq = np.asarray(q, np.int64)[:, None, :]
    qv = np.sign(cq) * (np.abs(cq) // qq)
    counts = np.zeros((2, 3, 256), np.int64)
    for c in range(3):
        dc = qv[c, :, 0]
        diff = np.abs(np.diff(np.concatenate([[0], dc])))
        nbits = np.where(diff > 0, np.floor(np.log2(np.maximum(diff, 1))).astype(np.int64) + 1, 0)
        np.add.at(counts[0, c], nbits, 1)
        zz = qv[c][:, ZIGZAG_NATURAL]
        for blk in zz:
            run = 0
            for k in range(1, 64):
                v = int(blk[k])
                if v == 0:
                    run += 1
                    continue
                while run > 15:
                    counts[1, c, 0xf0] += 1
                    run -= 16
                counts[1, c, (run << 4) + abs(v).bit_length()] += 1
                run = 0
            if run > 0:
                counts[1, c, 0] += 1
    return counts.astype(np.uint32)


def case_jpeg_entropy(L, host, w, h, x0=0, y0=0, check_histograms=True):
    """Device symbol statistics + device scan (gz_jpeg_histograms / gz_jpeg_scan) with the
    host-built marker segments must reproduce the reference's WriteJpeg byte for byte."""
    from checkers import ref
    rng = np.random.default_rng(RNG_SEED + 31 * w + h)
    rgb = images.crop(w, h, x0, y0) if max(w, h) <= 444 else images.tiled(w, h)
    co = oracle.encode_rgb(rgb)
    nb = co.shape[1]
    qs = [None,
          np.full((3, 64), 3, np.int32),
          np.stack([rng.integers(1, 9, 64), rng.integers(1, 30, 64), rng.integers(1, 30, 64)]).astype(np.int32),
          np.stack([rng.integers(200, 400, 64), rng.integers(1, 30, 64), rng.integers(1, 30, 64)]).astype(np.int32)]
    cases = []
    for q in qs:
        cq, _, _ = oracle.reconstruct(co, w, h, q)
        cases.append((cq, q, False))
    # dense large-magnitude coefficients (long MCUs, long codes, ZRL runs), q = 1
    wild = rng.integers(-2040, 2041, size=co.shape).astype(np.int16)
    wild[:, :, 1:][rng.random((3, nb, 63)) < 0.35] = 0
    wild[:, nb // 3:nb // 3 + 2, 1:40] = 0
    cases.append((wild, np.ones((3, 64), np.int32), False))
    # grey image: both chroma planes zero -> a single-component frame
    grey = cases[1][0].copy()
    grey[1:] = 0
    cases.append((grey, qs[1], True))
    with L.context(rgb, 1.0) as ctx:
        for cq, q, is_grey in cases:
            qq = np.ones((3, 64), np.int32) if q is None else q
            ctx.set_coeffs(cq)
            counts = ctx.jpeg_histograms(qq)
            if check_histograms:
                assert_bits_equal(counts, expected_jpeg_histograms(cq, qq), "jpeg histograms")
            ncomp = 1 if is_grey else 3
            head, depth, code = host.jpeg_head(counts, w, h, q, ncomp)
            n = ctx.jpeg_scan(ncomp, depth, code)
            scan = ctx.jpeg_scan_bytes()
            assert len(scan) == n
            # the scan's length in bits is a function of the symbol statistics and the code lengths
            # alone (what the search driver bounds a candidate's size with, without coding it);
            # the stuffed bytes are the 0xFF bytes of the stream
            bits, stuffed = ctx.jpeg_scan_bits()
            cnt = np.asarray(counts, np.int64)[:, :ncomp]
            extra = np.arange(256) & 15
            assert bits == int((cnt * (depth[:, :ncomp].astype(np.int64) + extra)).sum())
            assert n == (bits + 7) // 8 + stuffed and stuffed == scan.count(b"\xff\x00")
            got = head + scan + b"\xff\xd9"
            exp = host.write_jpeg(cq, w, h, q)           # pinned to the reference in
            assert got == exp, (len(got), len(exp))      # test_host_encoder.test_write_jpeg_bytes
            if ref is not None and q is not None and not is_grey and cq is not wild:
                assert got == ref.write_jpeg(co, w, h, q)
            ctx.jpeg_scan_keep()
            assert ctx.jpeg_scan_bytes(kept=True) == scan


# ------------------------------------------------------------------ YUV 4:2:0 (row f4) --
def colourful(w, h):
    """Saturated reds / blues over dark and bright backgrounds with soft and hard edges: makes
    every branch of PreProcessChannel (sharpen map, blur map, neither) non-trivial."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 60 + 150 * (np.sin(x / 9.0) > 0) * (y < 0.6 * h) + 40 * np.sin(y / 7.0)
    g = 40 + 30 * np.cos((x + y) / 13.0) + 120 * (x > 0.7 * w)
    b = 50 + 160 * (np.cos(y / 11.0) > 0.3) * (x < 0.5 * w) + 100 * (x > 0.7 * w)
    m = ((x - 0.5 * w) ** 2 + (y - 0.5 * h) ** 2) < (0.2 * min(w, h)) ** 2
    r[m], g[m], b[m] = 220, 30, 40
    return np.ascontiguousarray(np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8))


def case_frame420(L, w, h, chk, x0=0, y0=0, rgb=None):
    """gz_downsample (OutputImage::Downsample + PreProcessChannel), the 4:2:0 pixel model
    (gz_reconstruct), gz_compare and gz_block_weights_factor on a 4:2:0 frame against `chk`."""
    rng = np.random.default_rng(RNG_SEED + 11 * w + h)
    if rgb is None:
        rgb = images.crop(w, h, x0, y0)
    co = chk.encode_rgb(rgb)
    exp = chk.downsample(co, w, h)
    oc = chk.comparator(rgb, 1.0)
    with L.context(rgb, 1.0) as ctx:
        ctx.encode_rgb()
        got = ctx.downsample()
        assert ctx.frame_layout() == (2, ctx.nb, ctx.nbc)
        assert_bits_equal(got, exp, f"downsample {w}x{h}")
        q = np.stack([rng.integers(1, 8, 64), rng.integers(1, 12, 64),
                      rng.integers(1, 12, 64)]).astype(np.int32)
        for qq in (None, q):
            cq = ctx.quantize(qq)
            ecq, esrgb, elin = chk.reconstruct420(exp, w, h, qq, shuffle=7)
            assert_bits_equal(cq, ecq, "quantize (4:2:0)")
            srgb, lin = ctx.reconstruct()
            assert_bits_equal(srgb, esrgb, "reconstruct srgb (4:2:0)")
            assert_bits_equal(lin, elin, "reconstruct linear (4:2:0)")
        dist, dm, _ = ctx.compare()
        edist, edm = oc.compare420(cq)
        assert_bits_equal(dm, edm, "distance map (4:2:0)")
        assert dist == edist
        for direction in (1, -1):
            for factor in (1, 2):
                for radius in (1, 3):
                    gotw = ctx.block_weights_factor(direction, radius, 0.97, factor)
                    expw = oc.block_weights_factor(direction, radius, 0.97, factor, edm)
                    assert_bits_equal(gotw, expw, f"block weights factor {factor} dir {direction}")
        # back to 4:4:4
        ctx.encode_rgb()
        assert ctx.frame_layout()[0] == 1
    oc.close()


def case_block_search420(L, w, h, chk, x0=0, y0=0, qs=3, lookahead=3, new_model=True):
    """Phase A on a 4:2:0 frame: luma candidates on the 8x8 grid (mask 1), chroma candidates on
    the 16x16 grid with the 2x2 pixel model and the maximum over the sub-blocks (mask 6)."""
    rgb = images.crop(w, h, x0, y0)
    co = chk.encode_rgb(rgb)
    orig = chk.downsample(co, w, h)
    oc = chk.comparator(rgb, 0.971769)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb()
        ctx.downsample(download=False)
        cq = ctx.quantize(np.full((3, 64), qs, np.int32))
        for mask in (1, 6):
            off, idx, err = ctx.block_zeroing_orders(lookahead, new_model, comp_mask=mask)
            eoff, eidx, eerr = oc.block_zeroing_orders_masked(cq, orig, True, mask, lookahead, new_model)
            assert_bits_equal(off, eoff, f"offsets mask {mask}")
            assert_bits_equal(idx, eidx, f"candidates mask {mask}")
            assert_bits_equal(err, eerr, f"errors mask {mask}")
    oc.close()


def case_block_search_masks444(L, w, h, chk, x0=0, y0=0, qs=3, lookahead=3, new_model=True):
    """Component masks and non-default Params on a 4:4:4 frame."""
    rgb = images.crop(w, h, x0, y0)
    orig = chk.encode_rgb(rgb)
    oc = chk.comparator(rgb, 0.971769)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb()
        cq = ctx.quantize(np.full((3, 64), qs, np.int32))
        for mask in (7, 1, 6):
            off, idx, err = ctx.block_zeroing_orders(lookahead, new_model, comp_mask=mask)
            eoff, eidx, eerr = oc.block_zeroing_orders_masked(cq, orig, False, mask, lookahead, new_model)
            assert_bits_equal(off, eoff, f"offsets mask {mask}")
            assert_bits_equal(idx, eidx, f"candidates mask {mask}")
            assert_bits_equal(err, eerr, f"errors mask {mask}")
    oc.close()


def case_jpeg_entropy420(L, H, w, h, chk, x0=0, y0=0):
    """The device entropy coder on a 4:2:0 frame (MCUs of 2x2 luma + Cb + Cr blocks, padding
    blocks, DC prediction in scan order) + the host head == the reference's WriteJpeg."""
    rng = np.random.default_rng(RNG_SEED + 5 * w + h)
    rgb = images.crop(w, h, x0, y0)
    co = chk.encode_rgb(rgb)
    orig = chk.downsample(co, w, h)
    with L.context(rgb, 1.0) as ctx:
        ctx.encode_rgb()
        ctx.downsample(download=False)
        qs = [np.full((3, 64), 2, np.int32),
              np.stack([rng.integers(1, 9, 64), rng.integers(1, 30, 64), rng.integers(1, 30, 64)]).astype(np.int32),
              np.stack([np.full(64, 3), np.full(64, 4000), np.full(64, 4000)]).astype(np.int32)]   # chroma -> all zero
        for q in qs:
            cq = ctx.quantize(q)
            exp = chk.write_jpeg420(orig, w, h, q)
            assert H.write_jpeg(cq, w, h, q, factor=2) == exp, "host writer (4:2:0)"
            y, cb, cr = ctx.split420(cq)
            ncomp = 3 if (cb.any() or cr.any()) else 1
            counts = ctx.jpeg_histograms(q, ncomp)
            head, depth, code = H.jpeg_head(counts, w, h, q, ncomp=ncomp, factor=2)
            n = ctx.jpeg_scan(ncomp, depth, code)
            scan = ctx.jpeg_scan_bytes()
            assert len(scan) == n
            # the scan's length in bits is a function of the symbol statistics and the code lengths
            # alone (what the search driver bounds a candidate's size with, without coding it);
            # the stuffed bytes are the 0xFF bytes of the stream
            bits, stuffed = ctx.jpeg_scan_bits()
            cnt = np.asarray(counts, np.int64)[:, :ncomp]
            extra = np.arange(256) & 15
            assert bits == int((cnt * (depth[:, :ncomp].astype(np.int64) + extra)).sum())
            assert n == (bits + 7) // 8 + stuffed and stuffed == scan.count(b"\xff\x00")
            got = head + scan + b"\xff\xd9"
            assert got == exp, (len(got), len(exp), ncomp)


def case_global_order420(L, w, h, chk, x0=0, y0=0, qs=3, target=0.971769):
    """Phase B's order on the chroma grid of a 4:2:0 frame: device-side weights over 16x16
    areas, the construction loop, whole-block steps on components 1 and 2 with their
    statistics change."""
    rng = np.random.default_rng(RNG_SEED + 13 * w + h)
    rgb = images.crop(w, h, x0, y0)
    oc = chk.comparator(rgb, target)
    with L.context(rgb, target) as ctx:
        ctx.encode_rgb()
        orig = ctx.downsample()
        q = np.full((3, 64), qs, np.int32)
        cq = ctx.quantize(q)
        off, idx, err = ctx.block_zeroing_orders(comp_mask=6)
        dist, dm, _ = ctx.compare()
        gn = ctx.nbc
        cnt = np.diff(off)
        max_err = np.zeros(gn, np.float32)
        ctx.order_reset()
        for direction in (1, -1):
            next_cand = (rng.integers(0, 1000, gn) % (cnt + 1)).astype(np.int32)
            for radius in (1, 2):
                wgt = oc.block_weights_factor(direction, radius, 1.0, 2, dm)
                exp = []
                btc = 0
                for b in range(gn):
                    if wgt[b] == 0:
                        continue
                    e = err[off[b]:off[b + 1]]
                    at = next_cand[b]
                    if direction > 0:
                        vals = (e[at:] - max_err[b]) / wgt[b]
                        btc += at < cnt[b]
                    else:
                        vals = (max_err[b] - e[:at][::-1]) / wgt[b]
                        btc += at > 0
                    exp.extend((b, v) for v in vals.astype(np.float32))
                total, got_btc, _ = ctx.order_build_auto(direction, radius, 1.0, True, next_cand)
                assert total == len(exp) and got_btc == btc, (total, len(exp), got_btc, btc)
                got = ctx.order_fetch(0, total)
                if total:
                    assert_bits_equal(got["block"], np.array([b for b, _ in exp], np.int32), "order blocks")
                    assert_bits_equal(got["val"], np.array([v for _, v in exp], np.float32), "order vals")
        # whole-block steps
        def quantize(raw, qq):
            r = int(np.fmod(raw, qq))
            d = qq - r if 2 * r > qq else (-qq - r if -2 * r > qq else -r)
            return np.int16(raw + d)
        co = ctx.get_coeffs()
        coff = [0, ctx.nb, ctx.nb + ctx.nbc]
        for direction in (1, -1):
            exp = co.copy()
            next_cand = (rng.integers(0, 1000, gn) % (cnt + 1)).astype(np.int32)
            if direction > 0:
                counts = (rng.integers(0, 1000, gn) % (cnt - next_cand + 1)).astype(np.int32)
            else:
                counts = (rng.integers(0, 1000, gn) % (next_cand + 1)).astype(np.int32)
            sel = np.flatnonzero(counts > 0).astype(np.int32)
            for b in sel:
                for j in range(counts[b]):
                    p = next_cand[b] + j if direction > 0 else next_cand[b] - 1 - j
                    ix = int(idx[off[b] + p])
                    c, k = ix // 64, ix % 64
                    ob = orig[coff[c] + b].astype(np.int64)
                    newval = 0 if direction > 0 else int(quantize(int(ob[k]), qs))
                    precious = False
                    if newval == 0 and k in (1, 8):
                        hf = sum(abs(int(ob[i])) for i in range(3, 64) if not ((i & 7) < 3 and i < 24))
                        precious = abs(int(ob[k])) >= (4 if hf < 60 else 8)
                    if not precious:
                        exp[coff[c] + b, k] = newval
            ctx.order_build_auto(direction, 1, 1.0, True, next_cand)
            before = ctx.jpeg_histograms(q)
            ctx.apply_candidate_steps(direction, sel, counts[sel])
            delta = ctx.steps_histogram_delta()
            co = ctx.get_coeffs()
            assert_bits_equal(co, exp, f"apply_candidate_steps (4:2:0) direction {direction}")
            after = ctx.jpeg_histograms(q)
            assert_bits_equal(delta, after[1].astype(np.int64) - before[1].astype(np.int64),
                              f"steps_histogram_delta (4:2:0) direction {direction}")
    oc.close()


def case_compare_blocks(L, w, h, x0=0, y0=0, qs=3, n=24):
    """gz_compare_blocks == Comparator::SwitchBlock + CompareBlock (the per-block seam)."""
    rng = np.random.default_rng(RNG_SEED + 17 * w + h)
    rgb = images.crop(w, h, x0, y0)
    oc = oracle.comparator(rgb, 0.971769)
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb()
        cq = ctx.quantize(np.full((3, 64), qs, np.int32))
        bw = ctx.bw
        xy, blocks, exp = [], [], []
        for _ in range(n):
            b = int(rng.integers(0, ctx.nb))
            cand = cq.copy()
            for _ in range(int(rng.integers(0, 4))):   # zero a few coefficients of the block
                cand[int(rng.integers(0, 3)), b, int(rng.integers(1, 64))] = 0
            xy.append((b % bw, b // bw))
            blocks.append(cand[:, b, :])
            exp.append(oc.compare_block(cand, b % bw, b // bw))
        got = ctx.compare_blocks(np.array(xy), np.stack(blocks))
        assert_bits_equal(got, np.array(exp, np.float64), "CompareBlock")
        # the pixel form of the seam (gz_compare_block_pixels): the windows' YCbCr pixels are the
        # integer IDCT of their blocks, edge-replicated as OutputImageComponent::ToPixels does
        px = L.idct_blocks(np.stack(blocks).reshape(-1, 64)).reshape(-1, 3, 8, 8)
        for i, (bx, by) in enumerate(xy):
            vw, vh = min(8, w - 8 * bx), min(8, h - 8 * by)
            px[i, :, :, vw:] = px[i, :, :, vw - 1:vw]
            px[i, :, vh:, :] = px[i, :, vh - 1:vh, :]
        got_px = ctx.compare_block_pixels(np.array(xy), px.reshape(-1, 3, 64))
        assert_bits_equal(got_px, np.array(exp, np.float64), "CompareBlock (pixels)")
    oc.close()
