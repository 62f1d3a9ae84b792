"""Writes tests/golden/images.npz: small image files (made by tests/image_cases.py) together with the pixels the
reference's Texture::Texture (its vendored stb_image; src/Texture.cpp:4-8) decodes from them.  Needs oracle/_ref
(built from /root/reference by oracle/build_ref.sh); run from the repo root: python tests/golden/make_image_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from oracle import binding  # noqa: E402
import image_cases  # noqa: E402


def main():
    if not binding.have_ref():
        binding.build_ref()
    R = binding.Reference()
    rng = np.random.default_rng(11)
    cases = [c for c in image_cases.jpeg_cases(rng, keep=0.008) if len(c[1]) < 4000]
    rng = np.random.default_rng(12)
    cases += [c for c in image_cases.lossless_cases(rng) if "37x23" in c[0] or "5x9" in c[0]]
    rng = np.random.default_rng(13)
    cases += [c for c in image_cases.more_cases(rng) if ("37x23" in c[0] or "5x9" in c[0] or "ramp." in c[0]) and len(c[1]) < 6000]
    arrays, names = {}, []
    with tempfile.TemporaryDirectory() as d:
        for k, (name, data) in enumerate(cases):
            path = os.path.join(d, name)
            with open(path, "wb") as f:
                f.write(data)
            px = R.texture_pixels(path)
            assert px is not None, name
            arrays["file_%d" % k] = np.frombuffer(data, np.uint8)
            arrays["pixels_%d" % k] = px
            names.append(name)
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **arrays)
    print(len(names), "files,", os.path.getsize(os.path.join(HERE, "images.npz")), "bytes")


if __name__ == "__main__":
    main()
