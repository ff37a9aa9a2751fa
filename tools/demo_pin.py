#!/usr/bin/env python3
"""demo_pin.py — which std.rand variant / draw order reproduces the reference's images/demo.png?

The reference holds ONE rendered image (README.md:9: scene 21, 150 spp, depth 6, seed 456).  This
script (CPU, test infrastructure: it uses the oracle) builds the scene-21 sphere table under a set
of candidate generators and draw orders (scene.zig:57-74), renders each with the oracle at configs[0]'s
256x144 and reports the PSNR against the box-resampled demo.  A candidate that is the
reference's generator must beat every other seed / variant by a wide margin.

    python tools/demo_pin.py [--spp 16] [--write-fixture]

--write-fixture stores the demo box-resampled by 1/3 (data only: 223x398x3 uint8) as tests/golden/demo_398x223.npz;
tests/test_demo_pin.py renders at the demo's own width (1193: the picture was NOT made at the prompt's default 256 —
the v = (H - row + 1 + xi)/H quirk shifts a 144-row frame by 9 demo pixels, a 671-row frame by 2) and resamples
its frame with the same operator.
Reads /root/reference/images/demo.png, so it runs in the authoring container only.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

M64 = (1 << 64) - 1
DEMO = "/root/reference/images/demo.png"
W, H = 256, 144  # BASELINE configs[0]: the prompt's default width (main.zig:58); demo.png is 1193x669 = 4.66 x this


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


class SplitMix64:
    def __init__(self, s):
        self.s = s & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)


class Xoshiro256pp:  # Zig 0.9.x std/rand/Xoshiro256.zig
    def __init__(self, seed):
        g = SplitMix64(seed)
        self.s = [g.next() for _ in range(4)]

    def next(self):
        s = self.s
        r = (rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return r


class Xoroshiro128p:  # Zig <= 0.8 DefaultPrng (std/rand/Xoroshiro128.zig: 55, 14, 36)
    def __init__(self, seed, consts=(55, 14, 36)):
        g = SplitMix64(seed)
        self.s = [g.next(), g.next()]
        self.c = consts

    def next(self):
        s0, s1 = self.s
        r = (s0 + s1) & M64
        s1 ^= s0
        a, b, c = self.c
        self.s[0] = rotl(s0, a) ^ s1 ^ ((s1 << b) & M64)
        self.s[1] = rotl(s1, c)
        return r


class Pcg32:  # std/rand/Pcg.zig (not the default in any release; tried for completeness)
    def __init__(self, seed):
        g = SplitMix64(seed)
        self.s = 0
        self.i = (g.next() << 1 | 1) & M64  # seedTwo(gen.next(), gen.next()): i = (i<<1)|1 ; s = 0; s += init_s ...
        init_s = g.next()
        # Zig: seedTwo(init_s, init_i): self.s = 0; self.i = (init_s << 1) | 1; self.s = self.s *% default_multiplier +% self.i; self.s +%= init_i; ...
        # kept approximate: a mismatch here only means this candidate cannot match
        self.s = (self.s * 6364136223846793005 + self.i) & M64
        self.s = (self.s + init_s) & M64
        self.s = (self.s * 6364136223846793005 + self.i) & M64

    def next32(self):
        l = self.s
        self.s = (l * 6364136223846793005 + (self.i | 1)) & M64
        xor_s = (((l >> 18) ^ l) >> 27) & 0xFFFFFFFF
        rot = l >> 59
        return ((xor_s >> rot) | (xor_s << ((32 - rot) & 31))) & 0xFFFFFFFF

    def next(self):
        return self.next32()


def f32_from_u32(s):
    return np.uint32(0x3F800000 | (s >> 9)).view(np.float32) - np.float32(1.0)


def make_float(gen, which):
    if which == "lo32":
        return lambda: f32_from_u32(gen.next() & 0xFFFFFFFF)
    if which == "hi32":
        return lambda: f32_from_u32((gen.next() >> 32) & 0xFFFFFFFF)
    if which == "f64":  # a float(f64) narrowed to f32
        return lambda: np.float32((np.uint64(0x3FF0000000000000 | (gen.next() >> 12))).view(np.float64) - 1.0)
    raise ValueError(which)


def scene21(rnd, order="ref"):
    """scene.zig:52-84 with the draws taken from rnd(); `order` permutes the per-sphere draw order."""
    f = np.float32
    out = [(0.0, -1000.0, 0.0, 1000.0, 0, 0.5, 0.5, 0.5, 0.0)]
    a = -11
    while a < 11:
        b = -11
        while b < 11:
            if order == "ref":          # scene.zig:61-62
                choose = rnd(); cx = f(a) + f(0.9) * rnd(); cz = f(b) + f(0.9) * rnd()
            elif order == "center_first":
                cx = f(a) + f(0.9) * rnd(); cz = f(b) + f(0.9) * rnd(); choose = rnd()
            elif order == "z_first":    # right-to-left argument evaluation
                choose = rnd(); cz = f(b) + f(0.9) * rnd(); cx = f(a) + f(0.9) * rnd()
            else:
                raise ValueError(order)
            if choose < f(0.8):
                al = (rnd(), rnd(), rnd())
                out.append((cx, 0.2, cz, 0.2, 0, al[0], al[1], al[2], 0.0))
            elif choose < f(0.95):
                al = (rnd(), rnd(), rnd())
                fz = f(0.5) * rnd()
                out.append((cx, 0.2, cz, 0.2, 1, al[0], al[1], al[2], fz))
            else:
                out.append((cx, 0.2, cz, 0.2, 2, 1.0, 1.0, 1.0, 1.5))
            b += 1
        a += 1
    out.append((0.0, 1.0, 0.0, 1.0, 2, 1.0, 1.0, 1.0, 1.5))
    out.append((-4.0, 1.0, 0.0, 1.0, 0, 0.4, 0.2, 0.1, 0.0))
    out.append((4.0, 1.0, 0.0, 1.0, 1, 0.7, 0.6, 0.5, 0.0))
    sp = np.zeros(len(out), dtype=oracle.SPHERE_DTYPE)
    for i, t in enumerate(out):
        sp[i] = t
    return sp


FIX_W, FIX_H = 398, 223  # the fixture: demo.png (1193x669) box-resampled by ~1/3


def demo_resampled(W=W, H=H):
    """Box-resample demo.png (1193x669) onto a W x H pixel grid: each target pixel is the area-weighted mean of the
    source pixels it covers."""
    from PIL import Image
    src = np.asarray(Image.open(DEMO).convert("RGB"), dtype=np.float64)
    sh, sw, _ = src.shape

    def weights(n_src, n_dst):
        Wm = np.zeros((n_dst, n_src))
        scale = n_src / n_dst
        for i in range(n_dst):
            lo, hi = i * scale, (i + 1) * scale
            for j in range(int(np.floor(lo)), min(n_src, int(np.ceil(hi)))):
                Wm[i, j] = max(0.0, min(hi, j + 1) - max(lo, j))
            Wm[i] /= Wm[i].sum()
        return Wm

    Wy, Wx = weights(sh, H), weights(sw, W)
    return np.einsum("lk,ikc->ilc", Wx, np.einsum("ij,jkc->ikc", Wy, src))   # rows, then columns


def render(spheres, spp, seed=1):
    img = oracle.image_new(W)
    assert (img.width, img.height) == (W, H)
    cam = oracle.camera_preset(21)
    cfg = oracle.make_config(img, subpixels=4, rays_per_subsample=spp // 4, rebounds=6, seed=seed)
    rc, s, _ = oracle.render_sum(spheres, cam, cfg, nthreads=8)
    assert rc == 0
    mean = oracle.resolve_mean(cfg, s)
    return np.clip(mean * 255.0, 0, 255)


def psnr(a, b, region=None):
    if region is not None:
        a, b = a[region], b[region]
    mse = np.mean((a - b) ** 2)
    return 10.0 * np.log10(255.0 ** 2 / mse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--write-fixture", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="only write the fixture")
    args = ap.parse_args()
    oracle.build()
    demo = demo_resampled()
    if args.write_fixture:
        path = os.path.join(ROOT, "tests", "golden", "demo_398x223.npz")
        np.savez_compressed(path, rgb=np.clip(np.round(demo_resampled(FIX_W, FIX_H)), 0, 255).astype(np.uint8),
                            src_size=np.array([1193, 669]))
        print("wrote", path)
    if args.no_search:
        return
    ground = np.zeros((H, W), dtype=bool)
    ground[H // 3:, :] = True  # below the horizon: where the 484 small spheres are

    cands = []
    for seed in (456, 457, 0, 1):
        for eng_name, eng in (("xoshiro256++", Xoshiro256pp), ("xoroshiro128+", Xoroshiro128p)):
            for which in ("lo32", "hi32"):
                for order in ("ref", "center_first", "z_first"):
                    if seed != 456 and (order != "ref"):
                        continue
                    cands.append((seed, eng_name, eng, which, order))
    rows = []
    for seed, eng_name, eng, which, order in cands:
        sp = scene21(make_float(eng(seed), which), order)
        im = render(sp, args.spp)
        rows.append((psnr(im, demo), psnr(im, demo, ground), seed, eng_name, which, order))
        print("seed %3d %-14s %-5s %-13s  PSNR full %.2f dB  ground %.2f dB" % (seed, eng_name, which, order, rows[-1][0], rows[-1][1]),
              flush=True)
    # the oracle's own table as a cross-check of this script's xoshiro/lo32/ref candidate
    sp_o, _ = oracle.setup_scene(21, 456)
    sp_p = scene21(make_float(Xoshiro256pp(456), "lo32"), "ref")
    same = all(np.array_equal(sp_o[k], sp_p[k]) for k in sp_o.dtype.names)
    print("python xoshiro256++/lo32/ref table == oracle table:", same)


if __name__ == "__main__":
    main()
