"""Statistical comparison of candidate dropout hashes (numpy, CPU).  A hash maps (seed, pair index) -> 32 bits; element
2n takes the low 16-bit half, element 2n+1 the high half; kept when half >= thresh.  The candidates use only
operations that are full rate on gfx950 (add, xor, shifts, v_mad_u32_u24); the baseline is the murmur3 finaliser
with three 32-bit multiplies (quarter rate)."""
import sys, numpy as np
U = np.uint64; M32 = U(0xFFFFFFFF); M24 = U(0xFFFFFF)

def mad24(a, b, c): return ((a & M24) * (U(b) & M24) + c) & M32
def xs(x, k): return x ^ (x >> U(k))

def murmur(seed, n):
    x = (n * U(0x9E3779B1) + U(seed)) & M32
    x = xs(x, 16); x = (x * U(0x85EBCA6B)) & M32
    x = xs(x, 13); x = (x * U(0xC2B2AE35)) & M32
    return xs(x, 16)

def cand(c1, c2, s1, s2, s3, r1, r2):
    def f(seed, n):
        x = (n + U(seed)) & M32
        x = xs(x, s1)
        x = mad24(x, c1, x >> U(r1))
        x = xs(x, s2)
        x = mad24(x, c2, x >> U(r2))
        return xs(x, s3)
    return f

def cand3(c1, c2, c3, sh):
    def f(seed, n):
        x = (n + U(seed)) & M32
        for c, (s, r) in zip((c1, c2, c3), sh):
            x = xs(x, s)
            x = mad24(x, c, x >> U(r))
        return xs(x, 15)
    return f

def halves(h): return (h & U(0xFFFF)).astype(np.int64), (h >> U(16)).astype(np.int64)

def elems(f, seed, idx0, n_elem):
    """keep-field (16 bit) of elements idx0 .. idx0+n_elem (idx0 even)."""
    n = (np.arange(n_elem // 2, dtype=np.uint64) + U(idx0 // 2)) & M32
    lo, hi = halves(f(seed, n))
    out = np.empty(n_elem, np.int64); out[0::2] = lo; out[1::2] = hi
    return out

def corr(a, b):
    a = a - a.mean(); b = b - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))

def evaluate(name, f, N=1 << 24):
    res = {}
    seeds = [0x12345678, 0x9E3779B1, 1, 0xDEADBEEF]
    # 1. keep rate
    worst = 0.0
    for p in (0.1, 0.25, 0.5):
        th = round(p * 65536)
        for s in seeds:
            k = (elems(f, s, 0, N) >= th)
            z = abs(k.mean() - (1 - th / 65536)) / np.sqrt(p * (1 - p) / N)
            worst = max(worst, z)
    res["keep-rate worst |z|"] = worst
    # 2. serial correlation of the keep bit at strides that are tensor strides
    th = round(0.1 * 65536)
    k = (elems(f, seeds[0], 0, N) >= th).astype(np.float64)
    lags = [1, 2, 3, 4, 8, 16, 63, 64, 65, 128, 255, 256, 257, 512, 768, 769, 3072, 65536, 65536 * 12, 1 << 20, 1 << 22]
    res["serial worst |z|"] = max(abs(corr(k[:-l], k[l:])) * np.sqrt(N - l) for l in lags)
    # 2b. far strides (heads / batch entries: 2^16 .. 2^24 elements apart), windows of 2^20 elements
    w = 1 << 20
    base = (elems(f, seeds[0], 0, w) >= th).astype(np.float64)
    far = [1 << 24, (1 << 24) * 3, 1 << 25, (1 << 16) * 12 * 5, 1 << 28, (1 << 28) + (1 << 24), 1 << 30, (1 << 31) - (1 << 20) * 2]
    res["far-stride worst |z|"] = max(abs(corr(base, (elems(f, seeds[0], d, w) >= th).astype(np.float64))) * np.sqrt(w) for d in far)
    # 3. cross-seed correlation (seed patterns the host produces: +1, xor const, + multiples of the mixing constants)
    s0 = 0x3C6EF372
    others = [s0 + 1, s0 ^ 0x9E3779B1, (s0 + 0x85EBCA6B) & 0xFFFFFFFF, (s0 + 2 * 0x85EBCA6B) & 0xFFFFFFFF, (s0 + 0xC2B2AE35 * 9) & 0xFFFFFFFF, s0 + 2, s0 + (1 << 16), s0 + (1 << 24)]
    a = (elems(f, s0, 0, w) >= th).astype(np.float64)
    res["cross-seed worst |z|"] = max(abs(corr(a, (elems(f, s, 0, w) >= th).astype(np.float64))) * np.sqrt(w) for s in others)
    # 4. avalanche: flip input bit b (of the pair index, and of the seed) -> fraction of output bits flipped
    n = (np.arange(1 << 18, dtype=np.uint64) * U(2654435761) ) & M32     # scattered inputs
    h0 = f(seeds[0], n)
    worst_bit = 0.0
    for b in range(32):
        h1 = f(seeds[0], n ^ U(1 << b))
        d = h0 ^ h1
        for ob in range(32):
            fr = float(((d >> U(ob)) & U(1)).mean())
            worst_bit = max(worst_bit, abs(fr - 0.5))
    res["avalanche worst |flip-0.5| (index bits)"] = worst_bit
    n2 = np.arange(1 << 18, dtype=np.uint64)                            # consecutive inputs (the real use)
    h0 = f(seeds[0], n2)
    worst_bit = 0.0
    for b in range(32):
        h1 = f(seeds[0] ^ (1 << b), n2)
        d = h0 ^ h1
        for ob in range(32):
            worst_bit = max(worst_bit, abs(float(((d >> U(ob)) & U(1)).mean()) - 0.5))
    res["avalanche worst (seed bits, consecutive idx)"] = worst_bit
    # 5. collisions over 2^24 consecutive pair indices (random function: ~N^2/2^33 = 32768)
    h = f(seeds[1], np.arange(N, dtype=np.uint64))
    res["collisions in 2^24"] = int(N - np.unique(h).size)
    # 6. bit balance of the two halves
    res["bit balance worst |z|"] = max(abs(float(((h >> U(b)) & U(1)).mean()) - 0.5) * 2 * np.sqrt(N) for b in range(32))
    print(f"{name}")
    for k_, v in res.items():
        print(f"    {k_:48s} {v:.4g}")
    return res

if __name__ == "__main__" and len(sys.argv) == 1:
    evaluate("murmur3 finaliser (3 x v_mul_lo_u32)", murmur)
    evaluate("A: xs16 mad24(0xB5297A,>>9) xs13 mad24(0x68E31D,>>11) xs16", cand(0xB5297A, 0x68E31D, 16, 13, 16, 9, 11))
    evaluate("B: xs15 mad24(0xD35A2D,>>8) xs12 mad24(0x9E3779,>>10) xs15", cand(0xD35A2D, 0x9E3779, 15, 12, 15, 8, 10))
    evaluate("C: 3 rounds", cand3(0xB5297A, 0x68E31D, 0xD35A2D, ((16, 9), (13, 11), (14, 8))))


def rotr(x, k): return ((x >> U(k)) | (x << U(32 - k))) & M32

def cand_rot(consts, rots, final=15, pre=0):
    def f(seed, n):
        x = (n + U(seed)) & M32
        if pre:
            x = xs(x, pre)
        for c, k in zip(consts, rots):
            x = mad24(x, c, rotr(x, k))
        return xs(x, final) if final else x
    return f

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rot":
    evaluate("D: 3 x mad24(x, C, rotr(x,k)) k=11,13,9 + xs15   (9 ops)", cand_rot((0xD35A2D, 0x9E3779, 0xB5297B), (11, 13, 9)))
    evaluate("E: 4 x mad24(x, C, rotr(x,k)) k=11,13,9,15 no final (9 ops)", cand_rot((0xD35A2D, 0x9E3779, 0xB5297B, 0x68E31D), (11, 13, 9, 15), final=0))
    evaluate("F: 2 x mad24(x, C, rotr) k=13,11 + xs15           (7 ops)", cand_rot((0xD35A2D, 0x9E3779), (13, 11)))
    evaluate("G: xs16 + 2 x mad24(x, C, rotr) k=13,11 + xs15    (9 ops)", cand_rot((0xD35A2D, 0x9E3779), (13, 11), pre=16))


def stress(name, f, N=1 << 26):
    """Heavier battery for an adopted candidate."""
    print(name)
    th = round(0.1 * 65536)
    zs = []
    for s in (0x12345678, 0x9E3779B1, 1, 0xDEADBEEF, 0, 0xFFFFFFFF, 0x80000000, 0x0000FFFF):
        e = elems(f, s, 0, N)
        k = (e >= th).astype(np.float32)
        zs.append(abs(float(k.mean()) - (1 - th / 65536)) / np.sqrt(0.1 * 0.9 / N))
        within = abs(corr(k[0::2].astype(np.float64), k[1::2].astype(np.float64))) * np.sqrt(N / 2)
        lag = max(abs(corr(k[:-l].astype(np.float64), k[l:].astype(np.float64))) * np.sqrt(N - l) for l in (1, 2, 256, 257, 768, 65536))
        # joint distribution of the top 4 bits of consecutive elements' fields: chi-square, 255 dof
        a, b = (e[:-1] >> 12), (e[1:] >> 12)
        tab = np.bincount(a * 16 + b, minlength=256).astype(np.float64)
        exp = (N - 1) / 256.0
        chi = float(((tab - exp) ** 2 / exp).sum())
        print(f"    seed {s:#010x}: keep-rate |z| {zs[-1]:.2f}  within-pair |z| {within:.2f}  lags worst |z| {lag:.2f}  chi2(255) {chi:.1f}  (z = {(chi - 255) / np.sqrt(510):+.2f})")
    # element-index windows high in the 32-bit space and wrap-around
    for idx0 in (0xF0000000, 0xFFFFFFF0 - (1 << 22), 0x7FFFFFF0):
        e = elems(f, 0x2468ACE0, idx0 & ~1, 1 << 22)
        k = (e >= th)
        print(f"    window at {idx0:#x}: keep-rate |z| {abs(float(k.mean()) - (1 - th / 65536)) / np.sqrt(0.09 / (1 << 22)):.2f}")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stress":
    stress("murmur3 finaliser", murmur)
    stress("B: xs15 mad24(0xD35A2D,>>8) xs12 mad24(0x9E3779,>>10) xs15", cand(0xD35A2D, 0x9E3779, 15, 12, 15, 8, 10))
