"""Statistics of the dropout keep mask as a function of the pair hash (CPU, numpy; no GPU needed).

The kernels draw one 32-bit hash per element PAIR and compare each 16-bit half with a threshold
(openvivqa_amd/csrc/common.h: drop_pair_hash / drop_keep).  This script compares candidate hashes on what dropout needs:
keep rate, lag correlations along the flat index (neighbours, row strides 512 / 2048, one sample = 6400 * 4), uniformity
of both 16-bit fields, and independence of the masks of different keys.  z-values should look standard normal.

  python scripts/dropout_hash_stats.py

Round 2 result (28 keys, n = 2^22, 1288 statistics per hash): `fmix32(pair ^ key)` -- worst |z| 3.6, sd 1.05 -- is as
good as the previous `fmix32(pair * 0x9E3779B1 ^ key)` -- 3.4, 1.03 -- and saves one quarter-rate 32-bit multiply per
pair.  (Its masks for keys that differ by d are the same mask with pair indices XOR-ed by d, so the "cross-key agreement"
of keys a few bits apart counts every pair twice: sd sqrt(2) by construction, not a defect; real keys are fmix32 outputs
and differ in their high bits.)  Two rounds of xorshift + 24-bit multiply-add (all full-rate instructions) are
measurably weaker (z sd 1.10) and were not adopted.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mul24(a, c):
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(c & 0xFFFFFF)) & M32


def mul32(a, c):
    return (a * np.uint64(c)) & M32


def fmix32(h):
    h = h ^ (h >> np.uint64(16))
    h = mul32(h, 0x85EBCA6B)
    h = h ^ (h >> np.uint64(13))
    h = mul32(h, 0xC2B2AE35)
    return h ^ (h >> np.uint64(16))


def premultiplied(x, key):   # round 1 / early round 2
    return fmix32(mul32(x, 0x9E3779B1) ^ np.uint64(key))


def plain(x, key):           # current: drop_pair_hash
    return fmix32(x ^ np.uint64(key))


def mad24_two_rounds(x, key):  # rejected
    h = x ^ np.uint64(key)
    h = h ^ (h >> np.uint64(14))
    h = (mul24(h, 0xC97833) + (h >> np.uint64(8))) & M32
    h = h ^ (h >> np.uint64(9))
    h = (mul24(h, 0xEC26C5) + (h >> np.uint64(8))) & M32
    return h ^ (h >> np.uint64(15))


def fields(f, idx, key):
    h = f(idx >> np.uint64(1), key)
    return np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))


def main(n=1 << 22):
    rng = np.random.default_rng(5)
    keys = [int(k) for k in rng.integers(0, 1 << 32, 24)] + [0, 1, 0xFFFFFFFF, 0x80000000]
    idx = np.arange(n, dtype=np.uint64)
    lags = (1, 2, 3, 4, 5, 6, 7, 8, 16, 32, 64, 128, 256, 511, 512, 513, 1024, 2047, 2048, 2049, 4096, 6400 * 4)
    for name, f in (("fmix32(pair * golden ^ key)", premultiplied), ("fmix32(pair ^ key)", plain),
                    ("2 x (xorshift, mad24)", mad24_two_rounds)):
        zs, chis, cross = [], [], []
        for key in keys:
            fld = fields(f, idx, key)
            for p in (0.1, 0.3):
                k = (fld >= np.uint64(int(p * 65536 + 0.5))).astype(np.float64)
                m = k.mean()
                zs.append((m - (1 - p)) / np.sqrt(p * (1 - p) / n))
                for lag in lags:
                    zs.append(((k[:-lag] - m) * (k[lag:] - m)).mean() / (p * (1 - p)) * np.sqrt(n))
            h = f(idx[: n // 2], key)  # distinct pairs
            for fl in (h & np.uint64(0xFFFF), h >> np.uint64(16)):
                cnt = np.bincount((fl >> np.uint64(6)).astype(np.int64), minlength=1024)
                e = cnt.sum() / 1024
                chis.append(((cnt - e) ** 2 / e).sum() / 1023)
        for key in keys[:12]:
            a = fields(f, idx, key) >= np.uint64(6554)
            for dk in (1, 2, 0x100, 0x10000, 0x80000000):
                b = fields(f, idx, key ^ dk) >= np.uint64(6554)
                cross.append(((a == b).mean() - 0.82) / np.sqrt(0.82 * 0.18 / n))
        zs = np.array(zs)
        print(f"{name:30s} {len(zs)} z-values: worst |z| {np.abs(zs).max():.2f}, sd {zs.std():.3f};  field chi2/dof "
              f"{min(chis):.3f}-{max(chis):.3f};  cross-key agreement worst |z| {np.abs(cross).max():.2f}, sd {np.std(cross):.2f}")


if __name__ == "__main__":
    main()
