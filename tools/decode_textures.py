#!/usr/bin/env python3
"""Decode scene textures to P6 PPM sidecars ("<file>.ppm") for the C++ host loader.

Equivalent of `image::open(path)?.to_rgb8()` (src/core/texture.rs:38): 8-bit
greyscale is replicated to RGB, alpha is dropped.  The JPEG decoder here is
libjpeg (PIL), the reference used the `jpeg-decoder` crate; the two may differ
by +-1 level on some texels ("parity unpinned" for texel values).

usage: tools/decode_textures.py scenes/textures/*.png scenes/textures/*.jpg
"""
import sys
from PIL import Image


def main(paths):
    for p in paths:
        im = Image.open(p).convert("RGB")
        with open(p + ".ppm", "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % im.size)
            f.write(im.tobytes())
        print(p, "->", p + ".ppm", im.size)


if __name__ == "__main__":
    main(sys.argv[1:])
