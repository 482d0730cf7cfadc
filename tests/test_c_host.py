"""The drop-in boundary from a plain C host: tests/native/c_host.c is compiled with gcc as C99 against include/aic_hip.h and linked
with libaic_hip.so -- no HIP headers, no C++, no Python underneath. What it draws must be what the same calls draw through the
ctypes binding, and what the oracle draws. Without a GPU the program must be told AIC_ERR_NO_DEVICE (the CPU leg)."""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from all_is_cubes_amd import abi

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "native" / "c_host.c"


def build_c_host(tmp_path) -> Path:
    exe = tmp_path / "c_host"
    lib_dir = ROOT / "all_is_cubes_amd"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", f"-I{ROOT / 'include'}", str(SRC), f"-L{lib_dir}", "-laic_hip",
           f"-Wl,-rpath,{lib_dir}", "-o", str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return exe


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "AIC_KEEP_HW_QUEUES")}


def write_scene(path: Path, flat_space, options: abi.Options, frame: abi.FrameDesc, up_xyz, up_block, up_light) -> None:
    d, p = abi.Context._space_desc(flat_space)
    head = np.zeros(1, np.dtype([("lo", "<i4", (3,)), ("size", "<i4", (3,)), ("n_blocks", "<u4"), ("sky_kind", "<i4"), ("n_voxels", "<u8"), ("n_palette", "<u8"),
                                 ("sky", "<f4", (8, 3)), ("block_sky", "u1", (7, 4)), ("n_update", "<u4")], align=True))
    assert head.dtype.itemsize == 176
    head["lo"], head["size"] = p.lo, p.size
    head["n_blocks"], head["sky_kind"], head["n_voxels"], head["n_palette"] = len(p.blocks), p.sky_kind, p.voxels.size, len(p.palette)
    head["sky"] = p.sky
    head["block_sky"] = np.array([[d.block_sky[i][j] for j in range(4)] for i in range(7)], np.uint8)
    head["n_update"] = len(up_xyz)
    with open(path, "wb") as f:
        f.write(b"AICSCENE")
        f.write(head.tobytes())
        for a in (p.block_index, p.light, p.blocks, p.voxels, p.palette):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(bytes(options))
        f.write(bytes(frame))
        f.write(np.ascontiguousarray(up_xyz, np.int32).tobytes())
        f.write(np.ascontiguousarray(up_block, np.uint16).tobytes())
        f.write(np.ascontiguousarray(up_light, np.uint8).tobytes())


def test_c_host_compiles_as_c99_and_is_refused_loudly_without_a_device(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU leg runs the program for real")
    exe = build_c_host(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=clean_env())
    assert out.returncode == 2, (out.returncode, out.stdout, out.stderr)
    assert "AIC_ERR_NO_DEVICE" in out.stdout
    # the library's load-time default is in place before main() of a host that links it
    assert out.stdout.splitlines()[0] == "GPU_MAX_HW_QUEUES=8"


@pytest.mark.gpu
def test_c_host_draws_what_ctypes_and_the_oracle_draw(tmp_path):
    import oracle
    from all_is_cubes_amd import workloads as scenes
    from tests.test_gpu_parity import to_abi_options

    space = scenes.synthetic_space(n=20, resolution=8, n_blocks=8, seed=11)
    w, h = 160, 100
    opt = oracle.make_options()
    eye = (10.5, 16.5, 34.0)
    q = oracle.look_at_y_up(eye, (10.0, 6.0, 10.0))
    _, _, inv = oracle.camera_matrices(90.0, opt.view_distance, w / h, q, eye)
    frame = abi.Context.make_frame(w, h, world_inv=inv)
    # the update: two cubes in the camera's view get another block and a light texel of their own
    lo = np.asarray(space.lo)
    up_xyz = np.array([lo + (10, 13, 15), lo + (9, 13, 15)], np.int32)
    up_block = np.array([1, 6], np.uint16)  # an atom and a recursive block
    up_light = np.array([[90, 80, 70, 128], [60, 70, 80, 128]], np.uint8)
    scene = tmp_path / "scene.bin"
    write_scene(scene, space, to_abi_options(opt), frame, up_xyz, up_block, up_light)

    exe = build_c_host(tmp_path)
    result = tmp_path / "frames.bin"
    out = subprocess.run([str(exe), str(scene), str(result)], capture_output=True, text=True, timeout=300, env=clean_env())
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.splitlines()[0] == "GPU_MAX_HW_QUEUES=8"
    raw = result.read_bytes()
    fsz = w * h * 4 + ctypes.sizeof(abi.FrameInfo)
    assert len(raw) == 2 * fsz
    frames = []
    for k in range(2):
        img = np.frombuffer(raw, np.uint8, w * h * 4, k * fsz).reshape(h, w, 4)
        info = abi.FrameInfo.from_buffer_copy(raw, k * fsz + w * h * 4)
        frames.append((img, info))

    # the same calls through ctypes: bytes and step totals equal
    with abi.Context(0) as ctx:
        ctx.upload_space(abi.LAYER_WORLD, space)
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        first = ctx.render(frame)
        ctx.update_cubes(abi.LAYER_WORLD, up_xyz, up_block, up_light)
        second = ctx.render(frame)
    for (img, info), got in zip(frames, (first, second)):
        assert (img == got["rgba8"]).all()
        assert info.cubes_traced == got["info"].cubes_traced and info.rows_rendered == h
    assert (frames[0][0] != frames[1][0]).any(), "the update is in view"

    # and the oracle, before and after the update
    opt.exposure = 1.0
    for k in range(2):
        if k == 1:
            for (x, y, z), b, l in zip(up_xyz - lo, up_block, up_light):
                space.block_index[x, y, z] = b
                space.light[x, y, z] = l
        ref = oracle.render(oracle.Space(space), opt, oracle.make_camera(inv, w, h), threads=4)
        assert frames[k][1].cubes_traced == int(ref["info"]["cubes_traced"])
        assert np.abs(frames[k][0].astype(int) - ref["rgba8"].astype(int)).max() <= 1
