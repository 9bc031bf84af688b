"""Statistical checks of the attention dropout hash (csrc/attention.hip rng_mix) on the CPU: keep rate, correlations between neighbouring\nkeys / queries / heads / the two halves of a pair, bit balance and avalanche, for the 32-bit finalizer and the 24-bit-multiply one."""
import numpy as np
M32 = np.uint64(0xffffffff)
def u32(x): return (x & M32).astype(np.uint64)
def mul32(a, c): return u32(a * np.uint64(c))
def mul24(a, c): return u32((a & np.uint64(0xffffff)) * np.uint64(c & 0xffffff))
CQ, CK, CH = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D
def mix_old(x):
    x = x ^ (x >> np.uint64(16)); x = mul32(x, 0x7feb352d)
    x = x ^ (x >> np.uint64(15)); x = mul32(x, 0x846ca68b)
    return x ^ (x >> np.uint64(16))
def mix_new(x, c1=0xA2B5A5, c2=0xC6A4A7, s1=16, s2=13, s3=16):
    x = x ^ (x >> np.uint64(s1)); x = mul24(x, c1)
    x = x ^ (x >> np.uint64(s2)); x = mul24(x, c2)
    return x ^ (x >> np.uint64(s3))
def qpart(s1, bh, q): return u32(mul32(q, CQ) ^ u32(mul32(bh, CH) + np.uint64(s1)))
def kpart(s0, k): return u32(mul32(k, CK) + np.uint64(s0))

def decisions(mix, seed, nbh=8, nq=256, nk=256, p=0.3):
    s0, s1 = seed & 0xffffffff, seed >> 32
    bh = np.arange(nbh, dtype=np.uint64)[:, None, None]
    q = np.arange(nq, dtype=np.uint64)[None, :, None]
    kp = (np.arange(nk // 2, dtype=np.uint64) * np.uint64(2))[None, None, :]
    h = mix(qpart(s1, bh, q) ^ kpart(s0, kp))
    thr = np.uint64(int(p * 65536 + 0.5))
    lo = (h & np.uint64(0xffff)) >= thr
    hi = (h >> np.uint64(16)) >= thr
    keep = np.stack([lo, hi], -1).reshape(nbh, nq, nk)
    return keep, h

def report(name, mix):
    worst = {}
    for seed in [7, 991, 0x123456789abcdef, 2**63 + 12345, 1]:
        keep, h = decisions(mix, seed)
        k = keep.astype(np.float64)
        n = k.size
        rate = k.mean()
        z_rate = (rate - 0.7) / np.sqrt(0.21 / n)
        c = k - k.mean()
        def corr(a, b): return float((a * b).mean() / 0.21)
        stats = dict(
            z_rate=z_rate,
            z_adjk=corr(c[:, :, :-1], c[:, :, 1:]) * np.sqrt(c[:, :, 1:].size),
            z_pair=corr(c[:, :, 0::2], c[:, :, 1::2]) * np.sqrt(c[:, :, 0::2].size),
            z_adjq=corr(c[:, :-1, :], c[:, 1:, :]) * np.sqrt(c[:, 1:, :].size),
            z_adjh=corr(c[:-1], c[1:]) * np.sqrt(c[1:].size),
            z_diag=corr(c[:, :-1, :-1], c[:, 1:, 1:]) * np.sqrt(c[:, 1:, 1:].size),
            z_k2=corr(c[:, :, :-2], c[:, :, 2:]) * np.sqrt(c[:, :, 2:].size),
            z_q8=corr(c[:, :-8, :], c[:, 8:, :]) * np.sqrt(c[:, 8:, :].size),
            z_rowrate=float(((k.mean(2) - 0.7) / np.sqrt(0.21 / k.shape[2])).std()),   # ~1 for independent
            z_colrate=float(((k.mean(1) - 0.7) / np.sqrt(0.21 / k.shape[1])).std()),
        )
        # bit balance of h
        bits = ((h[..., None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)).astype(np.float64)
        stats["max_bit_bias_z"] = float(np.abs((bits.mean((0, 1, 2)) - 0.5) / np.sqrt(0.25 / h.size)).max())
        for kk, v in stats.items():
            worst[kk] = max(worst.get(kk, 0), abs(v)) if not kk.startswith("z_row") and not kk.startswith("z_col") else max(worst.get(kk, 0), abs(v - 1))
    print(name, {k: round(v, 2) for k, v in worst.items()})
    # avalanche: flip one input bit of x, fraction of output bits flipping
    rng = np.random.default_rng(0)
    x = rng.integers(0, 2**32, 200000, dtype=np.uint64)
    hx = mix(x)
    av = np.zeros((32, 32))
    for b in range(32):
        d = hx ^ mix(x ^ np.uint64(1 << b))
        av[b] = ((d[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)).mean(0)
    print("   avalanche: min %.3f max %.3f mean %.3f ; worst input bit mean %.3f ; worst output bit mean %.3f" % (av.min(), av.max(), av.mean(), av.mean(1).min(), av.mean(0).min()))

report("old(lowbias32)", mix_old)
report("new24(16,13,16)", mix_new)
report("new24(16,12,16)b", lambda x: mix_new(x, 0x9E3779, 0x85EBCB, 16, 12, 16))
report("new24(15,13,16)c", lambda x: mix_new(x, 0xB5297B, 0x68E31D, 15, 13, 16))
