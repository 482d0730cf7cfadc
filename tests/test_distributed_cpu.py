"""world_size-2 gloo test (CPU) of the N>1 path: interleaved row-strip partition, gather to
rank 0, de-interleave. The pixel source on CPU is the oracle (this is a test); on the GPU the
same functions are fed by the HIP renderer (tests/test_gpu_parity.py, bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from all_is_cubes_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, strip, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from tests import scenes
        from tests.test_oracle_goldens import camera_for

        sp = oracle.Space(scenes.synthetic_space(n=12, resolution=4, n_blocks=4, seed=6))
        eye = (6.5, 10.5, 20.0)
        cam = camera_for(w, h, eye, oracle.look_at_y_up(eye, (6, 4, 6)))
        rows = D.partition_rows(h, strip, world, rank)
        # each rank renders ONLY its rows (oracle renders row ranges; take the needed bands)
        local = np.zeros((len(rows), w, 4), np.uint8)
        i = 0
        while i < len(rows):
            j = i
            while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
                j += 1
            band = oracle.render(sp, oracle.make_options(), cam, rows=(rows[i], rows[j] + 1), threads=1)["rgba8"]
            local[i : j + 1] = band[rows[i] : rows[j] + 1]
            i = j + 1
        gathered = D.gather_strips(torch.from_numpy(local), h, w, strip)
        if rank == 0:
            full = D.assemble_strips_torch(gathered, h, w, strip).numpy()
            ref = oracle.render(sp, oracle.make_options(), cam, threads=1)["rgba8"]
            q.put(bool((full == ref).all()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,strip", [(70, 40, 16), (33, 17, 8)])
def test_two_rank_strip_gather(h, w, strip):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, h, w, strip, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_matches_abi_helper():
    import ctypes

    from all_is_cubes_amd import abi

    lib = abi.load()
    for h, strip, n in [(1080, 16, 8), (70, 16, 3), (5, 16, 2)]:
        for p in range(n):
            part = abi.Partition(strip, n, p, 0)
            assert lib.aic_partition_rows(h, ctypes.byref(part)) == len(D.partition_rows(h, strip, n, p))


def _pipeline_worker(rank, world, port, h, w, strip, frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def frame_image(f):  # what a full frame f looks like: every pixel names its frame and position
            y, x = np.mgrid[0:h, 0:w]
            return np.stack([(x + 3 * f) % 256, (y * 7 + f) % 256, (x ^ y) % 256, np.full_like(x, f)], -1).astype(np.uint8)

        pipe = D.StripGatherPipeline(h, w, strip, "cpu", depth=2)
        rows = D.partition_rows(h, strip, world, rank)
        done, ok = [], True

        def finish(slot):
            g = pipe.retire(slot)
            if rank == 0 and g is not None:
                done.append(D.assemble_strips_torch(g, h, w, strip).numpy().copy())

        for f in range(frames):
            slot = f % pipe.depth
            finish(slot)                                        # frame f-2 leaves the ring
            pipe.local[slot][: len(rows)] = torch.from_numpy(frame_image(f)[rows])  # "render" this rank's strips
            pipe.submit(slot)
        while pipe.oldest() is not None:
            finish(pipe.oldest())
        if rank == 0:
            ok = len(done) == frames and all((done[f] == frame_image(f)).all() for f in range(frames))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def _grouped_pipeline_worker(rank, world, port, h, w, strip, frames, per_gather, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def frame_image(f):
            y, x = np.mgrid[0:h, 0:w]
            return np.stack([(x + 3 * f) % 256, (y * 7 + f) % 256, (x ^ y) % 256, np.full_like(x, f)], -1).astype(np.uint8)

        pipe = D.StripGatherPipeline(h, w, strip, "cpu", depth=2, frames=per_gather)
        rows = D.partition_rows(h, strip, world, rank)
        done = []

        def finish(slot, n_valid):
            g = pipe.retire(slot)
            if rank == 0 and g is not None:
                for k in range(n_valid):
                    done.append(pipe.assemble(g, k).numpy().copy())

        valid = {}
        for f in range(frames):
            group, k = divmod(f, per_gather)
            slot = group % pipe.depth
            if k == 0:
                finish(slot, valid.pop(slot, 0))                # the group that last used this slot leaves the ring
            pipe.frame_buffer(slot, k)[: len(rows)] = torch.from_numpy(frame_image(f)[rows])
            valid[slot] = k + 1
            if k == per_gather - 1 or f == frames - 1:          # a full group, or the last, partial one
                pipe.submit(slot)
        while pipe.oldest() is not None:
            s_ = pipe.oldest()
            finish(s_, valid.pop(s_, 0))
        if rank == 0:
            q.put(bool(len(done) == frames and all((done[f] == frame_image(f)).all() for f in range(frames))))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_of_several_frames_per_collective():
    """`frames` > 1: one collective moves several consecutive frames; they come back whole and in order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    h, w, strip = 70, 12, 16
    port = _free_port()
    procs = [ctx.Process(target=_grouped_pipeline_worker, args=(r, 2, port, h, w, strip, 11, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_two_rank_pipelined_gather_keeps_frames_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 70, 24, 16, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
