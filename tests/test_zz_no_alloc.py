"""slam_process_frames_noalloc_dev (the batched call on the library's fixed device scratch) against
slam_process_frames_dev (the same call on the ctx workspace), through the C ABI.

Collected last: the file name sorts after every other test file.  Bars: every integer output (stats words, decoded
records, per-tile counts, corrected rows, per-block counts, cell colours, pixel maps, tiles) byte-equal; the two f64
outputs (mean matrix, blurred matrix) bit-equal too, since both entry points run the same kernels on the same
weights.  Two calls in a row on the fixed scratch, the second with a smaller input, so that anything the first call
left behind in the array would show.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _outputs(pipe, F):
    import torch
    torch.cuda.synchronize()
    st = pipe.stats_host().copy()
    out = dict(stats=st, tiles=pipe.tiles[:F].cpu().numpy().copy(),
               matrix=pipe.matrix.cpu().numpy().view(np.uint64).copy(), cells=pipe.compact[0].cpu().numpy().copy())
    rows, kept, blur = [], [], []
    for f in range(F):
        r, k = pipe.frame_rows(f, st), pipe.frame_kept(f, st)
        rows.append(np.stack([r[c].astype(np.int64) for c in ("flag", "ue", "bs", "rss", "clk")], 1))
        kept.append(np.stack([k[c].astype(np.int64) for c in ("ue", "bs", "rss", "clk")], 1))
        blur.append(pipe.frame_matrix(f, st)[1].view(np.uint64).copy())
    out.update(rows=rows, kept=kept, blur=blur,
               tile_cnt=pipe.tile_cnt.cpu().numpy().copy(), kept_cnt=pipe.kept_cnt.cpu().numpy().copy())
    return out


def _same(a, b, F, what):
    na = a["stats"].shape[0]
    assert np.array_equal(a["stats"], b["stats"]), what
    assert np.array_equal(a["tiles"], b["tiles"]), what
    for f in range(F):
        assert np.array_equal(a["rows"][f], b["rows"][f]), (what, f)
        assert np.array_equal(a["kept"][f], b["kept"][f]), (what, f)
        assert np.array_equal(a["blur"][f], b["blur"][f]), (what, f)
        assert a["rows"][f].shape[0] > 0 and a["kept"][f].shape[0] > 0, (what, f)
    assert na == F


def _run(B, N, F, first, tile, noalloc, sigma=1.0, use_log=False):
    from slamhip import synth
    from slamhip.pipeline import FramePipeline, default_mapping, synth_frames, viridis_lut
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    text = synth_frames(F, B, N, first_frame=first)
    pipe = FramePipeline(F, tl, tile_w=tile[0], tile_h=tile[1], mapping=default_mapping(), lut=viridis_lut(),
                         blur_sigma=sigma, use_log=use_log, reserve=not noalloc)
    pipe.set_uniform_layout(F, stride, tl)
    pipe.matrix.fill_(-1.0)
    pipe.blurred.fill_(-1.0)
    pipe.tiles.fill_(7)
    pipe.run(text, noalloc=noalloc)
    out = _outputs(pipe, F)
    return out, pipe


def test_noalloc_entry_point_matches_the_allocating_one_twice_in_a_row():
    shapes = [(64, 1024, 3, 11, (160, 128), 1.0, False),          # first call: three frames of 64 x 1024 records
              (16, 512, 2, 40, (96, 64), 2.0, True)]              # second call: smaller input, other blur, log scale
    want = [_run(B, N, F, first, tile, False, sg, lg)[0] for B, N, F, first, tile, sg, lg in shapes]
    got = [_run(B, N, F, first, tile, True, sg, lg)[0] for B, N, F, first, tile, sg, lg in shapes]
    for (B, N, F, *_), w, g in zip(shapes, want, got):
        _same(w, g, F, (B, N))
    # and once more the first shape after the smaller one: the array keeps the last call's contents
    again = _run(*shapes[0][:5], True, shapes[0][5], shapes[0][6])[0]
    _same(want[0], again, shapes[0][2], "again")


def test_noalloc_entry_point_takes_one_full_size_frame():
    """One 256 x 4096-record frame (34.6 MB of text) is within the fixed array."""
    w = _run(256, 4096, 1, 5, (256, 256), False)[0]
    g = _run(256, 4096, 1, 5, (256, 256), True)[0]
    _same(w, g, 1, "full size")


def test_noalloc_entry_point_rejects_what_does_not_fit():
    from slamhip import _lib, synth
    from slamhip.pipeline import FramePipeline, synth_frames
    B, N, F = 256, 4096, 4                                           # four full-size frames: beyond the fixed array
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    pipe = FramePipeline(F, tl, tile_w=64, tile_h=64, reserve=False)
    pipe.set_uniform_layout(F, stride, tl)
    text = synth_frames(F, B, N)
    with pytest.raises(_lib.SlamError) as e:
        pipe.run(text, noalloc=True)
    assert e.value.code == _lib.E_CAPACITY and "scratch" in e.value.msg
    # a blur wider than the by-value weights can carry
    _, small = _run(16, 512, 1, 0, (64, 64), True)
    small.bt.blur_sigma = 4.0
    with pytest.raises(_lib.SlamError) as e:
        small.ctx.check(small.ctx.lib.slam_process_frames_noalloc_dev(small.ctx.h, C.byref(small.bt), 7))
    assert e.value.code == _lib.E_RANGE


def _pipe_for(B, N, F, first, tile, sigma=1.0):
    from slamhip import synth
    from slamhip.pipeline import FramePipeline, default_mapping, synth_frames, viridis_lut
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    text = synth_frames(F, B, N, first_frame=first)
    pipe = FramePipeline(F, tl, tile_w=tile[0], tile_h=tile[1], mapping=default_mapping(), lut=viridis_lut(),
                         blur_sigma=sigma, reserve=False)
    pipe.set_uniform_layout(F, stride, tl)
    return pipe, text


def test_noalloc_calls_from_two_threads_are_ordered():
    """Two host threads, a ctx and a stream each, both on the fixed scratch at once: every call's outputs equal what
    the same call gives when it runs alone (the calls wait for each other on the device)."""
    import threading
    import torch
    shapes = [(64, 1024, 3, 11, (160, 128)), (48, 768, 2, 23, (96, 64))]
    alone = []
    for B, N, F, first, tile in shapes:
        pipe, text = _pipe_for(B, N, F, first, tile)
        pipe.run(text, noalloc=True)
        alone.append(_outputs(pipe, F))
        del pipe, text
    pipes = [_pipe_for(*s) for s in shapes]
    streams = [torch.cuda.Stream() for _ in shapes]
    torch.cuda.synchronize()
    rounds, errors = 12, []
    results = [[] for _ in shapes]
    barrier = threading.Barrier(len(shapes))

    def worker(i):
        try:
            pipe, text = pipes[i]
            F = shapes[i][2]
            barrier.wait()
            for r in range(rounds):
                with torch.cuda.stream(streams[i]):
                    pipe.tiles.fill_(7)
                    pipe.run(text, noalloc=True)
                    # the reads below run on the same stream, i.e. behind this call only: a later call of the other thread
                    # that did not wait for this one would show in them
                    snap = dict(stats=pipe.stats.clone(), tiles=pipe.tiles[:F].clone(), kept=pipe.k_bs.clone(),
                                kept_cnt=pipe.kept_cnt.clone(), cells=pipe.compact[0].clone())
                results[i].append(snap)
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(shapes))]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    assert not errors, errors
    for i, (B, N, F, first, tile) in enumerate(shapes):
        pipe = pipes[i][0]
        final = _outputs(pipe, F)
        _same(alone[i], final, F, ("threads, last round", i))
        for r, snap in enumerate(results[i]):
            assert np.array_equal(snap["tiles"].cpu().numpy(), alone[i]["tiles"]), (i, r)
            assert np.array_equal(snap["cells"].cpu().numpy(), alone[i]["cells"]), (i, r)
            assert np.array_equal(snap["kept_cnt"].cpu().numpy(), alone[i]["kept_cnt"]), (i, r)


def test_noalloc_entry_point_under_graph_capture():
    """The call captured into a HIP graph (nothing in it allocates, copies from the host or synchronises) and replayed
    on fresh text: same bytes as the direct call."""
    import torch
    B, N, F, tile = 32, 1024, 2, (128, 96)
    from slamhip import synth
    from slamhip.pipeline import synth_frames
    pipe, text = _pipe_for(B, N, F, 3, tile)
    pipe.run(text, noalloc=True)
    want_a = _outputs(pipe, F)
    other = synth_frames(F, B, N, first_frame=77)
    pipe.run(other, noalloc=True)
    want_b = _outputs(pipe, F)
    assert not np.array_equal(want_a["tiles"], want_b["tiles"])
    buf = text.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.run(buf, noalloc=True)                          # warm-up on the capturing stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        pipe.run(buf, noalloc=True)
    for src, want in ((text, want_a), (other, want_b), (text, want_a)):
        buf.copy_(src)
        pipe.tiles.fill_(9)
        g.replay()
        _same(want, _outputs(pipe, F), F, "replay")
