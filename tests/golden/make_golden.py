#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the reference checkout.

Run ONLY in the authoring container (needs /root/reference and Pillow); the outputs are
committed so that nothing on the GPU box ever reads /root/reference.

What is extracted (all of it is DATA held by the reference's own tests, no source text):

* ``png_<case>.npy``  -- decoded RGBA8 pixels of
  ``test-renderers/expected/renderers/<case>.png`` (the image-comparison goldens used by
  ``test-renderers/tests/ray-render.rs`` through ``cases/src/lib.rs``).
* ``ascii_print_space.txt`` / ``ascii_partial_voxels.txt`` -- the two 80x40 expected
  character frames asserted in ``all-is-cubes-render/src/raytracer/text.rs:216-258,297-340``.
* ``srgb_decode_lut.npy`` -- the 256-entry sRGB8 -> linear table
  (``all-is-cubes-base/src/math/color.rs`` ``CONST_SRGB_LOOKUP_TABLE``), needed to restate
  scenes whose blocks are declared as sRGB8 colours (``color_srgb_ramp``, ``emission``).
* ``font_system16.npy`` (+ ``all_is_cubes_amd/host/font_system16.inc``, the same table as a C array for the host mirror)
  -- the glyph bitmaps of ``all-is-cubes/src/text/font-system-7x16.png`` (Builtin::FontSystem16), which the info-text
  overlay is drawn with (renderer.rs:659-683).
* ``packed_light_lut.npy`` -- the 256-entry ``PACKED_LIGHT_SCALAR_LOOKUP_TABLE``
  (``all-is-cubes/src/space/light/data.rs:301-354``); the product *generates* its table from
  the defining formula and the test asserts equality with this fixture.
"""
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent

PNG_CASES = [
    "transparent_one-surf-all",
    "transparent_one-vol-all",
    "emission-all",
    "emission_only-surf-all",
    "emission_only-vol-all",
    "emission_semi-surf-all",
    "emission_semi-vol-all",
    "color_srgb_ramp-all",
    "viewport_prime-all",
    "layers_all-all",
    "layers_hidden_ui-all",
    "layers_ui_only-all",
    "no_character_but_ui-ray",
    # lighting scenes: need the light updater (oracle/aic_light.inc) to reproduce
    *[f"light_spread-{o}-all" for o in ("None", "Flat", "Coarse", "Linear", "Smoothstep")],
    *[f"light_on_slab-{o}-all" for o in ("None", "Flat", "Coarse", "Linear", "Smoothstep")],
    "fog-None-ray", "fog-Abrupt-all", "fog-Compromise-all", "fog-Physical-all",
    "tone_map-Clamp-1.0-0.5-all", "tone_map-Clamp-1.0-2.0-all", "tone_map-Reinhard-0.5-0.5-all",
    "tone_map-Reinhard-1.0-0.5-all", "tone_map-Reinhard-1.0-2.0-all",
    "template-light-bench-all",
    # round 2, second batch
    "debug_pixel_cost-ray",
    "furnace-Clear-Opaque-all", "furnace-Clear-Transparent-all", "furnace-Foggy-Opaque-all", "furnace-Foggy-Transparent-all",
    "bloom-0.0-all",
    "no_update-all", "no_update-2-all",
    "follow_options_change-all", "follow_options_change-2-all",
    "template-cornell-box-all",
    # compared under a mask (one block of the scene needs the text engine): see tests/test_oracle_goldens2.py
    "antialias-Always-ray", "antialias-None-all",
    *[f"sky-{f}-all" for f in ("NX", "NY", "NZ", "PX", "PY", "PZ")],
    "viewport_zero-all", "viewport_zero-2-all", "layers_none_but_text-all",
    # round 3: the info-text overlay (renderer.rs:659-683) drawn by the host mirror
    "info_text-1.0-all", "info_text-1.5-ray", "info_text-2.0-ray",
]


def floats_after(text: str, marker: str, count: int, wrapped: str = "") -> np.ndarray:
    start = text.index(marker)
    start = re.search(r"=\s*&?\[", text[start:]).end() + start
    end = text.index("];", start)
    body = text[start:end]
    if wrapped:
        vals = re.findall(re.escape(wrapped) + r"\(([^)]+)\)", body)
    else:
        vals = [v for v in re.split(r"[,\s]+", body) if v]
    arr = np.array([float(v) for v in vals], dtype=np.float32)
    assert arr.shape == (count,), arr.shape
    return arr


def ascii_frames() -> list[str]:
    src = (REF / "all-is-cubes-render/src/raytracer/text.rs").read_text()
    frames = []
    for m in re.finditer(r'assert_eq!\(\s*output,\s*"\\\n(.*?)\n\s*"\s*\);', src, re.S):
        lines = []
        for line in m.group(1).split("\n"):
            line = line.strip()
            assert line.endswith("\\n\\"), repr(line)
            lines.append(line[: -len("\\n\\")])
        assert len(lines) == 40 and all(len(l) == 80 for l in lines)
        frames.append("\n".join(lines) + "\n")
    assert len(frames) == 2
    return frames


def font_system16() -> np.ndarray:
    """The 192 glyph bitmaps of Builtin::FontSystem16 (all-is-cubes/src/text/font.rs:23-30: `font-system-7x16.png`, 16 glyphs
    per row, 7x16 cells) as [glyph][row] bytes, bit x of a byte = pixel x of the row; a pixel is set where the atlas has
    r > 0 && a > 0 (font.rs `rgba_to_bit`)."""
    from PIL import Image

    im = np.asarray(Image.open(REF / "all-is-cubes/src/text/font-system-7x16.png").convert("RGBA"), dtype=np.uint8)
    assert im.shape[1] == 7 * 16 and im.shape[0] % 16 == 0
    bits = (im[..., 0] > 0) & (im[..., 3] > 0)
    n = (im.shape[0] // 16) * 16
    out = np.zeros((n, 16), np.uint8)
    for g in range(n):
        cell = bits[(g // 16) * 16:(g // 16) * 16 + 16, (g % 16) * 7:(g % 16) * 7 + 7]
        out[g] = (cell * (1 << np.arange(7))).sum(axis=1)
    return out


def write_font_inc(glyphs: np.ndarray) -> None:
    """The product's copy of the font: a generated table in the host mirror's source tree (the overlay is drawn by the
    product's host code, so the data has to ship with it)."""
    path = OUT.parent.parent / "all_is_cubes_amd" / "host" / "font_system16.inc"
    lines = ["// GENERATED by tests/golden/make_golden.py from the reference's font atlas all-is-cubes/src/text/font-system-7x16.png",
             "// (Builtin::FontSystem16, text/font.rs:23-30): 192 glyphs x 16 rows, bit x of a byte = pixel x of the row (7 wide).",
             "// Data, not code: the pixels of a bitmap font. Glyph index = character - 0x20 (0x20..0x7f) or - 0x40 (0xa0..0xff).",
             f"static const unsigned char kFontSystem16[{glyphs.shape[0]}][16] = {{"]
    for g in glyphs:
        lines.append("    {" + ", ".join(f"0x{v:02x}" for v in g) + "},")
    lines.append("};")
    path.write_text("\n".join(lines) + "\n")


def main() -> int:
    from PIL import Image

    glyphs = font_system16()
    np.save(OUT / "font_system16.npy", glyphs)
    write_font_inc(glyphs)

    for case in PNG_CASES:
        im = Image.open(REF / "test-renderers/expected/renderers" / f"{case}.png").convert("RGBA")
        np.save(OUT / f"png_{case}.npy", np.asarray(im, dtype=np.uint8))

    color_rs = (REF / "all-is-cubes-base/src/math/color.rs").read_text()
    np.save(OUT / "srgb_decode_lut.npy", floats_after(color_rs, "static CONST_SRGB_LOOKUP_TABLE", 256))
    light_rs = (REF / "all-is-cubes/src/space/light/data.rs").read_text()
    np.save(
        OUT / "packed_light_lut.npy",
        floats_after(light_rs, "static PACKED_LIGHT_SCALAR_LOOKUP_TABLE", 256, wrapped="ps32"),
    )
    a, b = ascii_frames()
    (OUT / "ascii_print_space.txt").write_text(a)
    (OUT / "ascii_partial_voxels.txt").write_text(b)
    print("golden fixtures written to", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
