"""Call dumps (AIC_DUMP): the file format round-trips, the recorded calls rebuild
the scene the reference-side updates describe (updating.rs:107-172), and -- on the GPU -- a session
recorded by the library replays to the same frames."""
import os
from pathlib import Path

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat, replay
from tests import scenes


def _session(apply):
    """One scripted session: upload, cube + light deltas, block replacement / append, light volume,
    options, two frames. `apply` receives each call as (name, *args)."""
    sp = scenes.synthetic_space(n=14, resolution=8, n_blocks=5, seed=12, light="field")
    opt = abi.make_options(lighting=4, fog=2)
    w, h = 80, 56
    eye = (7.5, 13.5, 22.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (7, 5, 7)), eye)
    f0 = abi.Context.make_frame(w, h, world_inv=inv, backdrop=(0.1, 0.2, 0.3, 0.5))
    apply("upload_space", abi.LAYER_WORLD, sp)
    apply("set_options", abi.LAYER_WORLD, opt)
    apply("frame", f0)
    rng = np.random.default_rng(8)
    xyz = rng.integers(-1, 15, (60, 3)).astype(np.int32)  # some outside the space: ignored like the reference's scatter
    _, first = np.unique(xyz, axis=0, return_index=True)
    xyz = xyz[np.sort(first)]
    bi = rng.integers(0, len(sp.blocks), len(xyz)).astype(np.uint16)
    lt = np.stack([rng.integers(90, 200, len(xyz))] * 3 + [np.full(len(xyz), 255)], axis=1).astype(np.uint8)
    apply("update_cubes", abi.LAYER_WORLD, xyz, bi, lt)
    apply("replace_block", abi.LAYER_WORLD, 2, scenes.synthetic_blocks(4, 1, seed=40)[0])
    new_index = len(sp.blocks)
    apply("replace_block", abi.LAYER_WORLD, new_index, flat.atom((0.9, 0.1, 0.9, 1.0), emission=(0.2, 0.0, 0.0)))
    apply("update_cubes", abi.LAYER_WORLD, np.array([[3, 9, 3]], np.int32), np.array([new_index], np.uint16), None)
    light = sp.light.copy()
    light[..., 0:3] = np.minimum(light[..., 0:3].astype(int) + 5, 255).astype(np.uint8) * (light[..., 3:4] == 255)
    apply("update_light_volume", abi.LAYER_WORLD, light)
    eye2 = (2.5, 11.0, 20.0)
    _, _, inv2 = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye2, (7, 5, 7)), eye2)
    apply("frame", abi.Context.make_frame(w, h, world_inv=inv2))
    return sp, opt


def _to_oracle_options(o):
    return oracle.make_options(fog=int(o["fog"]), transparency=int(o["transparency"]), lighting=int(o["lighting"]),
                               view_distance=float(o["view_distance"]))


def _oracle_frames(path):
    """Every recorded frame, traced by the oracle from the scene state the records rebuild."""
    st, out = replay.SceneState(), []
    for r in replay.read_dump(path):
        if r.tag != replay.FRAME:
            st.apply(r)
            continue
        fr = r.data["frame"]
        sp = st.spaces[abi.LAYER_WORLD]
        cam = oracle.make_camera(np.array(fr["world"]["inverse_projection_view"]).reshape(4, 4), int(fr["width"]), int(fr["height"]))
        out.append(oracle.render(oracle.Space(sp), _to_oracle_options(st.options[abi.LAYER_WORLD]), cam, backdrop=tuple(fr["backdrop"]))["rgba8"])
    return out


def test_dump_round_trip_rebuilds_the_scene(tmp_path):
    path = tmp_path / "session.aic"
    direct = {}

    with replay.DumpWriter(path) as wtr:
        def apply(name, *args):
            getattr(wtr, name)(*args)
            # the same call applied by hand to a FlatSpace
            if name == "upload_space":
                direct["sp"] = args[1]
            elif name == "update_cubes":
                sp = direct["sp"]
                for k, (x, y, z) in enumerate(args[1] - np.array(sp.lo)):
                    if 0 <= x < sp.size[0] and 0 <= y < sp.size[1] and 0 <= z < sp.size[2]:
                        if args[2] is not None:
                            sp.block_index[x, y, z] = args[2][k]
                        if args[3] is not None:
                            sp.light[x, y, z] = args[3][k]
            elif name == "replace_block":
                sp = direct["sp"]
                if args[1] == len(sp.blocks):
                    sp.blocks.append(args[2])
                else:
                    sp.blocks[args[1]] = args[2]
            elif name == "update_light_volume":
                direct["sp"].light[...] = args[1]

        _session(apply)

    recs = list(replay.read_dump(path))
    assert [r.tag for r in recs] == [replay.UPLOAD, replay.OPTIONS, replay.FRAME, replay.CUBES, replay.BLOCK, replay.BLOCK, replay.CUBES,
                                     replay.LIGHT, replay.FRAME]
    st = replay.SceneState()
    for r in recs:
        st.apply(r)
    got, want = st.spaces[abi.LAYER_WORLD], direct["sp"]
    assert (got.block_index == want.block_index).all() and (got.light == want.light).all()
    assert len(got.blocks) == len(want.blocks)
    for a, b in zip(got.blocks, want.blocks):
        assert a.resolution == b.resolution and tuple(a.vlo) == tuple(b.vlo) and a.is_one == b.is_one
        assert (a.voxels == b.voxels).all() and (a.palette == b.palette).all()
    assert int(st.options[abi.LAYER_WORLD]["lighting"]) == 4 and int(st.options[abi.LAYER_WORLD]["fog"]) == 2
    frames = _oracle_frames(path)
    assert len(frames) == 2 and frames[0].shape == (56, 80, 4)
    with open(path, "rb") as f:
        blob = f.read()
    bad = tmp_path / "bad.aic"
    bad.write_bytes(blob[:-5])
    with pytest.raises(ValueError):
        list(replay.read_dump(bad))


@pytest.mark.gpu
def test_library_dump_replays_to_the_same_frames(tmp_path):
    path = tmp_path / "lib.aic"
    os.environ["AIC_DUMP"] = str(path)
    try:
        ctx = abi.Context(0)  # the recorder is armed when the context is created
    finally:
        del os.environ["AIC_DUMP"]
    live = []
    with ctx:
        def apply(name, *args):
            if name == "frame":
                live.append(ctx.render(args[0])["rgba8"].copy())
            else:
                getattr(ctx, name)(*args)

        _session(apply)
    again = replay.replay(path)
    assert len(again) == len(live) == 2
    for a, b in zip(again, live):
        assert (a == b).all()
    for a, b in zip(_oracle_frames(path), live):  # and the captured scene satisfies the usual parity bar
        assert np.abs(a.astype(np.int16) - b.astype(np.int16)).max() <= 1


# Any recording dropped under tests/golden/ (e.g. an Atrium or DemoCity session captured by the Rust shim with
# AIC_DUMP, rust/all-is-cubes-hip/README.md) is replayed and held to the usual parity bar -- no code change needed.
RECORDED = sorted((Path(__file__).resolve().parent / "golden").glob("*.aic"))


@pytest.mark.parametrize("path", RECORDED or [None], ids=[p.name for p in RECORDED] or ["no-recording"])
def test_recorded_sessions_parse_and_trace_on_the_oracle(path):
    if path is None:
        pytest.skip("no *.aic recording under tests/golden/ (none can be produced without the reference's toolchain)")
    frames = _oracle_frames(path)
    assert frames and all(f.ndim == 3 and f.shape[2] == 4 for f in frames)


@pytest.mark.gpu
@pytest.mark.parametrize("path", RECORDED or [None], ids=[p.name for p in RECORDED] or ["no-recording"])
def test_recorded_sessions_replay_on_the_device(path):
    if path is None:
        pytest.skip("no *.aic recording under tests/golden/")
    device = replay.replay(path)
    cpu = _oracle_frames(path)
    assert len(device) == len(cpu)
    for a, b in zip(cpu, device):
        assert np.abs(a.astype(np.int16) - b.astype(np.int16)).max() <= 1


def test_bench_builds_its_workload_from_a_recording(tmp_path):
    """bench.py --workload replay:<file>: the world space, size, options and camera matrix of the recording's last frame."""
    import bench

    path = tmp_path / "w.aic"
    with replay.DumpWriter(path) as wr:
        def apply(name, *args):
            if name == "frame":
                wr.frame(args[0])
            else:
                getattr(wr, name)(*args)
        sp, _ = _session(apply)
    space, size, eye, target, view_distance, label = bench.build_workload(f"replay:{path}")
    assert size == (80, 56) and eye is None and target is None and view_distance == 200.0 and "w.aic" in label
    assert space.block_index.shape == sp.block_index.shape
    frames = [r for r in replay.read_dump(path) if r.tag == replay.FRAME]
    assert bench.REPLAY["inv"] == [float(v) for v in frames[-1].data["frame"]["world"]["inverse_projection_view"]]
    assert int(bench.REPLAY["options"]["lighting"]) == 4 and int(bench.REPLAY["options"]["fog"]) == 2
