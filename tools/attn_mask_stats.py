"""Statistics of the four-decisions-per-hash attention dropout mask (csrc/common.h Ds6gKeep4Base), numpy restatement, CPU only:
keep rate of each of the four 16-bit decisions, their pairwise correlation, lag correlations along keys and queries, and
the variance of per-row / per-column keep counts against the binomial value.    python tools/attn_mask_stats.py [nbh]"""
import sys
import numpy as np


def words(pre):
    x = pre.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15)
    w0 = x * np.uint32(0x846ca68b); w0 ^= w0 >> np.uint32(16)
    w1 = x * np.uint32(0xC2B2AE35); w1 ^= w1 >> np.uint32(16)
    return w0, w1


def main():
    nbh = int(sys.argv[1]) if len(sys.argv) > 1 else 48          # (batch, head) pairs: 48 = the bs 12 step
    T, p = 962, 0.1
    Tq4 = (T + 3) // 4
    thr = int(float(np.float32(p)) * 4294967296.0) >> 16
    for seed, off in ((0x5DEECE66D, (3 << 40) + 177 * 1024), (0xdeadbeefcafe1234, 1 << 40)):
        key = np.uint32((seed & 0xffffffff) ^ (((seed >> 32) * 0x85ebca6b) & 0xffffffff))
        q = np.arange(nbh * T * Tq4, dtype=np.uint64) + np.uint64(off)
        pre = (q & np.uint64(0xffffffff)).astype(np.uint32) ^ key ^ ((q >> np.uint64(32)).astype(np.uint32) * np.uint32(0x9E3779B9))
        w0, w1 = words(pre)
        f = np.stack([w0 & np.uint32(0xffff), w0 >> np.uint32(16), w1 & np.uint32(0xffff), w1 >> np.uint32(16)], 0)
        keep = (f >= thr).astype(np.float32)
        n = keep.shape[1]
        print(f"seed {seed:#x}: {n} quads, keep rates {keep.mean(1, dtype=np.float64).round(5)} (expected {1 - thr / 65536:.5f}), "
              f"1 sigma of a correlation = {1 / np.sqrt(n):.1e}")
        print("  pairwise correlation of the four decisions:\n", np.corrcoef(keep).round(5))
        m = keep.reshape(4, nbh, T, Tq4).transpose(1, 2, 3, 0).reshape(nbh, T, Tq4 * 4)[:, :, :T]

        def corr(a, b):
            a = a.ravel() - a.mean(); b = b.ravel() - b.mean()
            return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
        print("  key lag 1 / 2 / 4 / 32:", [round(corr(m[:, :, :-l], m[:, :, l:]), 5) for l in (1, 2, 4, 32)],
              " query lag 1 / 2 / 4:", [round(corr(m[:, :-l], m[:, l:]), 5) for l in (1, 2, 4)])
        rs, cs = m.sum(2), m.sum(1)
        print(f"  per-row keep count var {rs.var():.1f}, per-column {cs.var():.1f}, binomial {T * 0.9 * 0.1:.1f}")


if __name__ == "__main__":
    main()
