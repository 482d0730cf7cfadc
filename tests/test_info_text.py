"""The info-text overlay (renderer.rs:659-683): the host mirror's draw_info_text -- a restatement of
FontDef::draw_str_monospaced with Builtin::FontSystem16 (all-is-cubes/src/text/font.rs:178-203, layout.rs:101-265) -- against
the reference's info_text goldens for the raytracer (test-renderers/cases/src/lib.rs:667-711). The scene of that case is an empty space
with a uniform sky, so the whole golden is one background colour plus the overlay: no tracing is needed to compare it,
pixel for pixel (COLOR_ROUNDING_MAX_DIFF is not even used)."""
import numpy as np
import pytest

from all_is_cubes_amd import _host as H

# cases/src/lib.rs:680-692 (the string literal's `\\` line continuations strip the leading spaces of every line)
INFO_TEXT = ("/\\/\\/\\/\\/\\/\\/\\/\\\n"
             "| Hello·world. |\n"
             "| Info text    |\n"
             "| test Nº 1.   |\n"
             "+--------------+--------\n"
             "¦\n¦\n¦\n¦\n¦\n")


# (at scale factors 1.5 and 2.0 the raytracer has its own expected images, `-ray`: it does not scale the text with the viewport
#  -- renderer.rs:665 "TODO: We should scale text" -- while the `-all` images of those factors belong to the GPU renderer)
@pytest.mark.parametrize("name", ["info_text-1.0-all", "info_text-1.5-ray", "info_text-2.0-ray"])
def test_info_text_golden(golden_dir, name):
    ref = np.load(golden_dir / f"png_{name}.npy")
    img = np.empty_like(ref)
    img[...] = ref[-1, -1]  # the sky: Rgb(1.0, 0.5, 0.0) encoded (255, 188, 0, 255)
    assert tuple(ref[-1, -1]) == (255, 188, 0, 255)
    H.draw_info_text(img, INFO_TEXT)
    assert (img == ref).all(), np.argwhere((img != ref).any(axis=-1))[:10]


def test_info_text_characters_and_clipping():
    img = np.zeros((30, 40, 4), np.uint8)
    H.draw_info_text(img, "", (1, 1, 1, 255), (9, 9, 9, 255))
    assert not img.any()
    # unavailable characters become '?' (font.rs:214-229), curly quotes the straight ones; text past the edge is dropped
    a, b = np.zeros((40, 200, 4), np.uint8), np.zeros((40, 200, 4), np.uint8)
    H.draw_info_text(a, "世’” x" + "W" * 60)
    H.draw_info_text(b, "?'\" x" + "W" * 60)
    assert (a == b).all() and a.any() and a[:, -1].any()
    # a glyph's outline is drawn after the previous glyph and may cover its edge: every set pixel is one of the two paints
    vals = {tuple(v) for v in a.reshape(-1, 4)}
    assert vals == {(0, 0, 0, 0), (0, 0, 0, 255), (255, 255, 255, 255)}
