"""The pixels in which the pinned oracle's bunny image differs from the reference's golden PPM (Test/CTESTtest/data/bunny.ppm, kept as
tests/golden/ref_bunny.ppm), classified: a silhouette flip (one image lit, the other black: the two closest-hit queries disagree on
hit / miss at a triangle edge -- Embree's own edge rule, which no fixture pins) or a +-1 byte step (both lit: the last bit of a float
-- Embree's rcp + Newton step instead of a true division -- crossing a truncation boundary of (uchar)(c * 255)).  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import scenes
from oracle import orc
from tests.conftest import GOLDEN, read_ppm
from tests.helpers import oracle_render

for name, builder in (("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)):
    fb, _ = oracle_render(builder(), 1)
    img = orc.fb_to_ppm_bytes(fb).astype(np.int64)
    gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % name)).astype(np.int64)
    d = img - gold
    ys, xs = np.nonzero(np.abs(d).sum(axis=2))
    print("%s: %d differing pixels, sum |byte diff| = %d (CTest tolerance 300)" % (name, len(ys), np.abs(d).sum()))
    for y, x in zip(ys, xs):
        o, g = img[y, x], gold[y, x]
        lit_o, lit_g = bool(o.any()), bool(g.any())
        kind = "silhouette flip (oracle %s, golden %s)" % ("lit" if lit_o else "black", "lit" if lit_g else "black") if lit_o != lit_g else (
            "+-1 byte step" if np.abs(o - g).max() <= 1 else "shading difference")
        print("  pixel (x=%d, y=%d)  oracle %s  golden %s  -> %s" % (x, y, tuple(int(v) for v in o), tuple(int(v) for v in g), kind))
