"""MI355X-native replacement for the all-is-cubes CPU raytracer path.

Layers:
  libaic_hip.so   -- hand-written HIP kernels + the C ABI (include/aic_hip.h)
  _host           -- C++ mirror of the reference's renderer interface above the C ABI
                     (HeadlessRenderer / HipRtRenderer / Camera / GraphicsOptions / Space ...)
  abi, flat       -- ctypes binding of the C ABI and the flat scene container (numpy)

Nothing here falls back to a CPU renderer; without the HIP library or a GPU the renderer
constructors raise.
"""
from . import flat  # noqa: F401

__all__ = ["flat", "abi", "host", "space_from_flat"]


def _torch_first():
    """One HIP runtime per process (see abi.load): PyTorch-ROCm brings its own libamdhip64; loaded before it, libaic_hip.so brings a second."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def __getattr__(name):
    import importlib

    if name in ("abi", "distributed"):
        return importlib.import_module("." + name, __name__)
    if name in ("host", "_host"):
        _torch_first()
        try:
            return importlib.import_module("._host", __name__)
        except ImportError as e:  # pragma: no cover
            raise ImportError("all_is_cubes_amd._host is not built: run __graft_entry__.build()") from e
    raise AttributeError(name)


def space_from_flat(flat_space):
    """Builds a `_host.Space` (the C++ mirror of `Space`) from a `flat.FlatSpace`."""
    _torch_first()
    from . import _host as H

    sp = H.Space(tuple(int(v) for v in flat_space.lo), tuple(int(v) for v in flat_space.size))
    if flat_space.sky_kind == 0:
        sp.sky.set_uniform(tuple(float(v) for v in flat_space.sky[0]))
    else:
        sp.sky.set_octants(flat_space.sky)
    for b in flat_space.blocks:
        sp.add_block(evoxels_from_blockdef(b))
    sp.load_contents(flat_space.block_index, flat_space.light)
    return sp


def evoxels_from_blockdef(b):
    from . import _host as H

    if b.is_one:
        e = H.Evoxels.from_one(tuple(float(v) for v in b.palette[0, 0:4]), tuple(float(v) for v in b.palette[0, 4:7]))
    else:
        e = H.Evoxels.paletted(b.resolution, tuple(int(v) for v in b.vlo), b.voxels, b.palette)
    e.is_air = bool(b.is_air)
    e.display_name = "" if b.name == "#" else b.name
    return e
