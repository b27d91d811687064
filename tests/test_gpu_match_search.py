"""The two match-search kernels (candidate lists with fallback / exhaustive) give identical results.

Both are checked per position against the oracle's best match (lz77.h:36-100 restated), on content
that keeps tiles on the candidate-list path, content that forces the fallback, and mixtures that
switch inside one image (tile = 2048 coded positions).  Runs of equal pixels, aligned run ends,
runs cut by the window start and by tile borders are the cases the candidate-list kernel reasons about."""
import numpy as np
import pytest

import _oracle as O
import webp_amd

pytestmark = pytest.mark.gpu


def run(img, w, h, exhaustive):
    import torch
    enc = webp_amd.Encoder()
    enc.set_match_search(int(exhaustive))
    d_in = torch.from_numpy(img.view(np.int32)).cuda()
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off, lens, status = enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap)
    assert status[0] == 0
    m = enc.fetch_stage(0, webp_amd.STAGE_MATCH)
    res = enc.fetch_stage(0, webp_amd.STAGE_RESIDUAL)
    dense, total = enc.last_dense_tiles()
    run.kinds = enc.last_tile_kinds()
    data = bytes(d_out[int(off[0]): int(off[0]) + int(lens[0])].cpu().numpy())
    enc.close()
    return m, res, dense, total, data


def check(img, w, h, expect=None, behind_the_lists=True, third_mode=3):
    """behind_the_lists: a third run in which the candidate lists refuse every tile, so that the forms behind them
    (frequent value, grams, then exhaustive) search all of it; its tile kinds are left in run.kinds."""
    m1, _, dense1, total1, s1 = run(img, w, h, 1)
    m0, res, dense, total, s0 = run(img, w, h, 0)
    kinds0 = run.kinds
    assert dense1 == total1 == total
    ln, ds = O.lz77_best_all(res)
    results = [m0, m1]
    if behind_the_lists:
        m2, _, _, _, s2 = run(img, w, h, third_mode)
        assert s2 == s0
        results.append(m2)
    else:
        run.kinds = kinds0
    # the search along the parse (palette images) leaves positions no parse can reach at "no match": every position is the
    # oracle's or untouched, and every position a parse from one of a tile's first 128 positions gets to is the oracle's
    on_a_chain = np.zeros(ln.size, dtype=bool)
    adv = np.maximum(ln, 1).astype(np.int64)
    for start in range(0, ln.size, 2048):
        for e in range(start, min(start + (128 if start else 1), ln.size)):
            p = e
            while p < min(start + 2048, ln.size) and not on_a_chain[p]:
                on_a_chain[p] = True
                p += adv[p]
    for k, m in enumerate(results):
        same = ((m & 0xFF) == ln) & (np.where(ln > 0, (m >> 8) & 0x7FF, 0) == ds)
        if k == 1:   # the exhaustive kernel alone
            assert same.all()
        assert (same | (m == 0)).all()
        assert same[on_a_chain].all()
    assert s0 == s1
    rc, ref = O.encode(img, w, h)
    assert rc == 0 and s0 == ref
    if expect == "sparse":
        assert dense == 0, (dense, total)
    elif expect == "dense":
        assert dense == total, (dense, total)
    elif expect == "mixed":
        assert 0 < dense < total, (dense, total)
    return dense, total


def test_photo_like_stays_on_candidate_lists():
    check(O.gen_image(3, 256, 96, 11), 256, 96, "sparse")


def test_flat_and_runs_stay_on_candidate_lists():
    # equal pixels in a row are one candidate run, however long
    check(np.full(300 * 40, 0xFF336699, dtype=np.uint32), 300, 40, "sparse")
    check(O.gen_image(4, 640, 40, 3), 640, 40, "sparse")
    rng = np.random.default_rng(3)
    runs = np.repeat(rng.integers(0, 1 << 24, size=900, dtype=np.uint32) | np.uint32(0xFF000000), rng.integers(1, 40, size=900))
    check(runs[:16000], 16000, 1, "sparse")
    check(runs[:16000], 125, 128, "sparse")


def test_two_colour_noise_needs_exhaustive():
    rng = np.random.default_rng(4)
    img = np.where(rng.random(300 * 40) < 0.5, np.uint32(0xFF336699), np.uint32(0xFF000000)).astype(np.uint32)
    check(img, 300, 40)                      # two colours: palette path, 8 pixels bundled per coded pixel
    img[-300:] = np.arange(300, dtype=np.uint32) * np.uint32(0x010305) | np.uint32(0xFF000000)  # more than 256 colours
    dense, total = check(img, 300, 40)
    assert run.kinds.get("lists", 0) <= 1, run.kinds  # (none of it is the candidate lists' in any mode)
    check(O.gen_image(5, 256, 64, 3, 2), 256, 64, None)


@pytest.mark.parametrize("ncolors", [2, 4, 16])
def test_bundled_palettes_are_searched_along_the_parse(ncolors):
    w, h = 1024, 96
    check(O.gen_image(5, w, h, 7, ncolors), w, h, third_mode=4)
    assert run.kinds.get("parse_walk", 0) > 0, run.kinds


def test_palette_walk_on_flat_periodic_and_noisy_bytes():
    rng = np.random.default_rng(77)
    pal = (np.arange(16, dtype=np.uint32) * np.uint32(0x00111111)) | np.uint32(0xFF000000)
    w, h = 2048, 24
    flat = np.full(w * h, pal[3], dtype=np.uint32)
    flat[5000:5003] = pal[1]; flat[30000] = pal[2]
    for img in (flat,
                pal[np.tile(rng.integers(0, 16, size=37), w * h // 37 + 1)[: w * h]],        # period 37 pixels
                pal[rng.integers(0, 2, size=w * h)],                                         # two colours, noise: 8 to a byte
                pal[np.repeat(rng.integers(0, 4, size=w * h // 5 + 1), 5)[: w * h]],         # four colours in runs of five
                pal[(rng.random(w * h) < 0.02).astype(np.int64) * rng.integers(1, 16, size=w * h)]):  # one colour with specks
        check(img.astype(np.uint32), w, h, third_mode=4)


def test_switch_inside_one_image():
    rng = np.random.default_rng(17)
    w, h = 512, 48                      # 24576 positions = 12 tiles
    img = rng.integers(0, 1 << 24, size=w * h, dtype=np.uint32) | np.uint32(0xFF000000)
    img[5000:9000] = np.where(rng.random(4000) < 0.5, np.uint32(0xFF101010), np.uint32(0xFF202020))  # noise of two colours across tile borders
    img[11000:12500] = 0xFF303030        # a flat stretch (one run)
    img[15000:15100] = img[14000:14100]  # long match inside a sparse tile, source in the tile before
    img[20480:20480 + 128] = img[20480 - 1024:20480 - 1024 + 128]  # distance 1024 exactly at a tile start
    check(img, w, h, "mixed")


def test_sources_in_previous_tile_and_tail():
    rng = np.random.default_rng(23)
    n = 2048 * 3 + 77
    a = rng.integers(0, 1 << 24, size=n, dtype=np.uint32) | np.uint32(0xFF000000)
    a[2048:2048 + 200] = a[2048 - 1024:2048 - 1024 + 200]   # first position of a tile, farthest source
    a[4090:4200] = a[3990:4100]                              # match running over a tile border
    a[-60:] = a[-160:-100]                                   # match ending at the last pixel (kept a literal there)
    check(a, n, 1, "sparse")
    check(a, 77, (n // 77), None)


@pytest.mark.parametrize("alpha", [2, 4, 16, 64, 256, 4096])
def test_small_alphabets_around_the_budget(alpha):
    # duplicate density from "every pixel repeats" to "almost none": crosses the candidate-list work budget
    rng = np.random.default_rng(alpha)
    w, h = 640, 40
    img = (rng.integers(0, alpha, size=w * h).astype(np.uint32) * np.uint32(0x00010203)) | np.uint32(0xFF000000)
    check(img, w, h)


def test_periodic_patterns():
    for period in (1, 2, 3, 7, 16, 100, 1000, 1024, 1025):
        rng = np.random.default_rng(period)
        base = rng.integers(0, 1 << 24, size=period, dtype=np.uint32) | np.uint32(0xFF000000)
        img = np.tile(base, 9000 // period + 1)[:9000]
        check(img, 9000, 1)


def test_graphics_generator():
    check(O.gen_image(4, 640, 64, 9), 640, 64)


def test_runs_aligned_and_cut_by_the_window():
    rng = np.random.default_rng(31)
    n = 2048 * 4
    a = rng.integers(0, 1 << 24, size=n, dtype=np.uint32) | np.uint32(0xFF000000)
    v, u = np.uint32(0xFF010203), np.uint32(0xFF0A0B0C)
    # a source run of 50, later a run of 20 followed by the same pixels that follow the source run:
    # the best source is the one whose run remainder is 20 (aligned), not the run start
    a[1000:1050] = v
    a[1050:1060] = a[3000:3010]
    a[1500:1520] = v
    a[1520:1530] = a[3000:3010]
    # a run longer than the window: every position inside is cut by the window start
    a[3500:5200] = u
    # runs of 128 and more (length cap), and a run that ends exactly at a tile border
    a[5300:5428] = v
    a[5500:5700] = v
    a[6100:6144] = u
    a[6144:6150] = a[3000:3006]
    # same value, many short runs with different successors
    for k in range(30):
        a[6300 + 40 * k: 6300 + 40 * k + 1 + (k % 7)] = v
    check(a, n, 1)
    check(a, 128, n // 128)
    check(a[:8001], 127, 63)


@pytest.mark.parametrize("seed", range(6))
def test_random_run_structures(seed):
    # few values, geometric run lengths: many equal-valued runs with all sorts of alignments
    rng = np.random.default_rng(100 + seed)
    nv = int(rng.choice([2, 3, 5, 17]))
    vals = rng.integers(0, 1 << 24, size=nv, dtype=np.uint32) | np.uint32(0xFF000000)
    pieces = []
    total = 0
    mean = float(rng.choice([1.5, 4, 20, 150]))
    while total < 12000:
        r = int(rng.geometric(1.0 / mean))
        pieces.append(np.full(r, vals[int(rng.integers(0, nv))], dtype=np.uint32))
        total += r
    img = np.concatenate(pieces)[:12000]
    check(img, 12000, 1)
    check(img, 100, 120)


@pytest.mark.parametrize("seed", range(3))
def test_matches_ending_just_past_the_length_limit(seed):
    # Copies of one long pattern, each spoilt at a few places: at most positions several distances match for 128 pixels
    # and more and end at different places above the limit, where the farthest must win and not the one that goes on
    # longest; flat stretches in between give matches that their own distance cuts.
    rng = np.random.default_rng(500 + seed)
    n = 14000
    period = int(rng.choice([131, 200, 517]))
    base = rng.integers(0, 1 << 24, size=period, dtype=np.uint32) | np.uint32(0xFF000000)
    a = np.tile(base, n // period + 1)[:n].copy()
    for p in rng.integers(0, n, size=n // 90):
        a[p] ^= np.uint32(1 + int(rng.integers(0, 255)))
    for k in range(12):
        s = int(rng.integers(0, n - 400))
        a[s: s + int(rng.integers(2, 300))] = np.uint32(0xFF000000 | int(rng.integers(0, 1 << 24)))
    check(a, n, 1)
    check(a, 125, n // 125)


@pytest.mark.parametrize("m", [1023, 1024, 1025, 1040, 1151, 2047, 2048, 2049, 3071, 3073])
def test_sizes_around_the_head_of_an_image(m):
    # The first 1024 positions of an image, where distances reach before the first pixel, are searched by a kernel of
    # their own in the exhaustive form; images that end just before, at and just behind that border and a tile's end.
    rng = np.random.default_rng(m)
    vals = rng.integers(0, 1 << 24, size=300, dtype=np.uint32) | np.uint32(0xFF000000)
    img = vals[rng.integers(0, 300, size=m)]
    img[: m // 2] = np.repeat(img[: m // 2: 3], 3)[: m // 2]  # short runs in the first half
    img[m // 3: m // 3 + 40] = img[0:40]                      # a copy of the image's first pixels
    check(img, m, 1)


@pytest.mark.parametrize("seed", range(3))
def test_few_frequent_values_without_runs(seed):
    # A skewed alphabet (a few values hold a few per cent of the pixels each) and hardly any equal neighbours: tiles stay
    # on the candidate lists with full buckets, whose entries are then kept group by group and walked from either end.
    rng = np.random.default_rng(900 + seed)
    nv = 700
    vals = rng.integers(0, 1 << 24, size=nv, dtype=np.uint32) | np.uint32(0xFF000000)
    wts = 1.0 / (np.arange(nv) + 6.0)
    idx = rng.choice(nv, size=15000, p=wts / wts.sum())
    same = np.flatnonzero(idx[1:] == idx[:-1]) + 1
    idx[same] = (idx[same] + 1 + rng.integers(0, 5, size=same.size)) % nv  # break up most runs
    img = vals[idx]
    img[9000:9300] = img[2000:2300]  # long matches far back, across a group border
    img[12000:12040] = img[11500:11540]
    dense, total = check(img, 15000, 1)
    assert dense < total, (dense, total)
    check(img, 125, 120)


def _frequent_value_image(rng, n, v0, mean_run, alphabet, pairs=0.1):
    """Runs of v0 (geometric lengths) with single pixels -- sometimes two or three -- of other values between them."""
    vals = rng.integers(0, 1 << 24, size=alphabet, dtype=np.uint32) | np.uint32(0xFF000000)
    out = []
    total = 0
    while total < n:
        r = int(rng.geometric(1.0 / mean_run))
        out.append(np.full(r, v0, dtype=np.uint32))
        k = 1 + int(rng.random() < pairs) + int(rng.random() < pairs / 3)
        o = vals[rng.integers(0, alphabet, size=k)]
        if rng.random() < 0.05:
            o[:] = o[0]  # a short run of another value
        out.append(o)
        total += r + k
    return np.concatenate(out)[:n]


@pytest.mark.parametrize("seed", range(8))
def test_one_frequent_value(seed):
    # What the predictor leaves of flat pictures: one value in runs, other pixels singly between them.  Such tiles leave
    # the candidate lists and are searched around the frequent value (k_lz77_dominant), not exhaustively.
    rng = np.random.default_rng(700 + seed)
    n = 2048 * 5 + int(rng.integers(0, 2048))
    v0 = np.uint32(0xFF000000) if seed % 2 == 0 else np.uint32(0xFF000000 | int(rng.integers(0, 1 << 24)))
    mean_run = float([6, 2.5, 15, 40, 6, 9, 3, 80][seed])
    alphabet = int([6, 3, 40, 1000, 300, 2, 12, 5][seed])
    img = _frequent_value_image(rng, n, v0, mean_run, alphabet)
    # matches that go on behind a run: copies of stretches that hold several runs
    for _ in range(12):
        s = int(rng.integers(1100, n - 400)); d = int(rng.integers(1, 1025)); l = int(rng.integers(5, 300))
        img[s:s + l] = img[s - d:s - d + l].copy()
    if seed == 3:
        img[4000:4200] = v0       # a run of 128 and more
        img[6000:7300] = v0       # longer than the window
    if seed == 4:
        img[-50:] = v0            # the image ends inside a run
    img[200:600] = (np.arange(400, dtype=np.uint32) * np.uint32(0x010305)) | np.uint32(0xFF000000)  # more than 256 colours: no palette
    check(img, n, 1, behind_the_lists=True)
    if alphabet >= 40:  # (with few other values the anchors' lists outgrow their pool and the tile goes on to the exhaustive search)
        assert run.kinds.get("frequent_value", 0) >= 3, run.kinds
    w = 125
    check(img[: (n // w) * w], w, n // w, behind_the_lists=True)


def test_frequent_value_tiles_next_to_others():
    rng = np.random.default_rng(801)
    n = 2048 * 8
    img = rng.integers(0, 1 << 24, size=n, dtype=np.uint32) | np.uint32(0xFF000000)
    img[3000:9000] = _frequent_value_image(rng, 6000, np.uint32(0xFF000000), 7.0, 200)
    img[9000:11000] = np.where(rng.random(2000) < 0.5, np.uint32(0xFF101010), np.uint32(0xFF202020))  # no frequent value: exhaustive
    img[12000:14500] = _frequent_value_image(rng, 2500, np.uint32(0xFF7F7F7F), 4.0, 60)
    check(img, n, 1, behind_the_lists=True)
    assert run.kinds.get("frequent_value", 0) >= 2 and run.kinds.get("exhaustive", 0) >= 1, run.kinds


@pytest.mark.parametrize("ncolors,w,h", [(2, 2048, 24), (4, 1024, 40), (16, 512, 64), (3, 517, 33), (2, 64, 700)])
def test_bundled_palettes_are_searched_by_grams(ncolors, w, h):
    # palette indices bundled into words: few values that repeat everywhere, matches of 128 words next to ones of two or three
    img = O.gen_image(5, w, h, 777, ncolors)
    check(img, w, h)
    assert run.kinds.get("grams", 0) >= 1, run.kinds


@pytest.mark.parametrize("seed", range(4))
def test_few_values_everywhere(seed):
    # words from a small alphabet with copies at all distances, spoilt here and there: long and short matches mixed, many sources per word
    rng = np.random.default_rng(1200 + seed)
    n = 2048 * 4 + 300
    alphabet = int([3, 9, 40, 120][seed])
    vals = rng.integers(0, 1 << 24, size=alphabet, dtype=np.uint32) | np.uint32(0xFF000000)
    if seed == 1:
        vals[0] = np.uint32(0xFFFFFFFF)   # the mark of an empty table slot is a value like any other
    period = int(rng.integers(5, 300))
    base = vals[rng.integers(0, alphabet, size=period)]
    img = np.tile(base, n // period + 1)[:n].copy()
    spoil = rng.integers(0, n, size=n // 12)
    img[spoil] = vals[rng.integers(0, alphabet, size=spoil.size)]
    img[5000:5400] = vals[rng.integers(0, alphabet, size=400)]          # noise of the same alphabet
    img[100:500] = (np.arange(400, dtype=np.uint32) * np.uint32(0x010305)) | np.uint32(0xFF000000)  # more than 256 colours: no palette
    check(img, n, 1)
    assert run.kinds.get("grams", 0) >= 2, run.kinds
    check(img[: (n // 97) * 97], 97, n // 97)
