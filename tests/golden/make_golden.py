"""Re-derives the analytic known-answer vectors in kat.json with plain Python/numpy (no oracle,
no product code) and checks that the committed JSON agrees.  Run: python tests/golden/make_golden.py"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def xoshiro256pp(s, n):
    M = (1 << 64) - 1
    rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & M
    out = []
    s = list(s)
    for _ in range(n):
        out.append((rotl((s[0] + s[3]) & M, 23) + s[0]) & M)
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
        s[3] = rotl(s[3], 45)
    return out


def get_uv(n):  # hitable.rs:65-71 in float32
    pi = f32(math.pi)
    theta = f32(np.arccos(f32(-f32(n[1]))))
    phi = f32(np.arctan2(f32(-f32(n[2])), f32(n[0]))) + pi
    return [float(f32(phi / (f32(2) * pi))), float(f32(theta / pi))]


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    assert [str(v) for v in xoshiro256pp([1, 2, 3, 4], 10)] == kat["xoshiro256pp_state_1_2_3_4"]
    for case in kat["get_uv"]:
        got = get_uv(case["n"])
        assert np.allclose(got, case["uv"], atol=1e-7), (case, got)
    # offset_hit_point((1,0,0),(1,0,0)): bits(1.0) + 256
    bits = np.array([1.0], dtype=np.float32).view(np.uint32)[0] + 256
    assert hex(int(bits)) == kat["offset_hit_point"]["out_bits_x"]
    assert np.array([bits], dtype=np.uint32).view(np.float32)[0] == kat["offset_hit_point"]["out"][0]
    # reflectance(1, 1.5) = r0 = ((1-1.5)/(1+1.5))^2 = 0.04 ; reflectance(0, 1.5) = r0 + (1-r0) = 1
    r0 = f32(f32(1 - 1.5) / f32(1 + 1.5)) ** 2
    assert abs(float(r0) - 0.04) < 1e-7
    print("kat.json agrees with the analytic derivations")


if __name__ == "__main__":
    main()
