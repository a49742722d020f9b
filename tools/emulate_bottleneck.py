"""CPU emulation of bottleneck64_kernel's data flow (csrc/rtn_bottleneck.hip), lane by lane: the permuted + swizzled weight images
in LDS, the transposed 16x16x32 products (weights = A operand, pixels = columns), accumulators of one product packed in place as
the B operand of the next, the 16-byte loads / stores of 8 consecutive channels.  Small-integer data, exact in float32; compared
with the three convolutions done directly.   python tools/emulate_bottleneck.py  -> max |error| 0"""
import numpy as np

IMG = 64 * 128


def perm_row(rho):
    f, q, r = rho >> 4, (rho >> 2) & 3, rho & 3
    return 32 * (f >> 1) + 8 * q + 4 * (f & 1) + r


def run(B, H, W, tail, seed):
    rng = np.random.RandomState(seed)
    M = B * H * W
    ain = rng.randint(-2, 3, size=(M, 64)).astype(np.float32)
    xin = rng.randint(-3, 4, size=(M, 256)).astype(np.float32)
    w2b = rng.randint(-1, 2, size=(64, 9, 64)).astype(np.float32)
    w2c = rng.randint(-1, 2, size=(256, 64)).astype(np.float32)
    w2a = rng.randint(-1, 2, size=(64, 256)).astype(np.float32)
    b2b, b2c, b2a = [rng.randint(-2, 3, size=n).astype(np.float32) for n in (64, 256, 64)]
    # ---- LDS images (elements instead of bytes: 64 per row, 8-element chunks)
    lds = np.zeros((17, 64, 64), np.float32)
    for im in range(17):
        for rho in range(64):
            src = perm_row(rho)
            for slot in range(8):
                chunk = slot ^ (rho & 7)
                if im < 9:
                    v = w2b[src, im, chunk * 8:chunk * 8 + 8]
                elif im < 13:
                    v = w2c[(im - 9) * 64 + src, chunk * 8:chunk * 8 + 8]
                else:
                    v = w2a[src, (im - 13) * 64 + chunk * 8:(im - 13) * 64 + chunk * 8 + 8]
                lds[im, rho, slot * 8:slot * 8 + 8] = v

    def wfrag(im, f, ks):              # [64 lanes][8]: lane (q, c) reads row 16 f + c, chunk (4 ks + q) at slot chunk ^ (c & 7)
        out = np.zeros((64, 8), np.float32)
        for lane in range(64):
            c, q = lane & 15, lane >> 4
            slot = (4 * ks + q) ^ (c & 7)
            out[lane] = lds[im, 16 * f + c, slot * 8:slot * 8 + 8]
        return out

    def mfma(acc, a, b):               # acc [64 lanes][4]; A lane l: A[row l&15][8 (l>>4) + e]; B lane l: B[8 (l>>4) + e][col l&15]
        A = np.zeros((16, 32), np.float32)
        Bm = np.zeros((32, 16), np.float32)
        for l in range(64):
            A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
            Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
        D = A @ Bm
        for l in range(64):
            for r in range(4):
                acc[l, r] += D[4 * (l >> 4) + r, l & 15]

    def bias_frag(bias, base, f):
        out = np.zeros((64, 4), np.float32)
        for lane in range(64):
            q = lane >> 4
            o = base + 32 * (f >> 1) + 8 * q + 4 * (f & 1)
            out[lane] = bias[o:o + 4]
        return out

    xout = np.full((M, 256), np.nan, np.float32)
    aout = np.full((M, 64), np.nan, np.float32)
    ball = np.concatenate([b2b, b2c, b2a])
    for strip in range((M + 31) // 32):
        p0 = strip * 32
        acc1 = [[bias_frag(ball, 0, f).copy() for u in range(2)] for f in range(4)]
        for t in range(9):
            dy, dx = t // 3 - 1, t % 3 - 1
            xb = np.zeros((2, 2, 64, 8), np.float32)   # [ks][u][lane]
            for u in range(2):
                for lane in range(64):
                    c, q = lane & 15, lane >> 4
                    pix = p0 + 16 * u + c
                    if pix >= M:
                        continue
                    rem = pix % (H * W)
                    y, x = rem // W, rem % W
                    if 0 <= y + dy < H and 0 <= x + dx < W:
                        for ks in range(2):
                            xb[ks, u, lane] = ain[pix + dy * W + dx, 32 * ks + 8 * q:32 * ks + 8 * q + 8]
            for ks in range(2):
                for f in range(4):
                    wf = wfrag(t, f, ks)
                    for u in range(2):
                        mfma(acc1[f][u], wf, xb[ks, u])
        h1 = np.zeros((2, 2, 64, 8), np.float32)       # [s][u]
        for s in range(2):
            for u in range(2):
                h1[s, u, :, 0:4] = np.maximum(acc1[2 * s][u], 0)
                h1[s, u, :, 4:8] = np.maximum(acc1[2 * s + 1][u], 0)
        acc3 = [[bias_frag(ball, 320, f).copy() for u in range(2)] for f in range(4)]
        for g in range(4):
            acc2 = [[bias_frag(ball, 64 + 64 * g, f).copy() for u in range(2)] for f in range(4)]
            for ks in range(2):
                for f in range(4):
                    wf = wfrag(9 + g, f, ks)
                    for u in range(2):
                        mfma(acc2[f][u], wf, h1[ks, u])
            xo = np.zeros((2, 2, 64, 8), np.float32)
            for s in range(2):
                for u in range(2):
                    for lane in range(64):
                        c, q = lane & 15, lane >> 4
                        pix = p0 + 16 * u + c
                        if pix >= M:
                            continue
                        ch = 64 * g + 32 * s + 8 * q
                        v = np.concatenate([acc2[2 * s][u][lane], acc2[2 * s + 1][u][lane]]) + xin[pix, ch:ch + 8]
                        v = np.maximum(v, 0)
                        xo[s, u, lane] = v
                        xout[pix, ch:ch + 8] = v
            if tail:
                for ks in range(2):
                    for f in range(4):
                        wf = wfrag(13 + g, f, ks)
                        for u in range(2):
                            mfma(acc3[f][u], wf, xo[ks, u])
        if tail:
            for s in range(2):
                for u in range(2):
                    for lane in range(64):
                        c, q = lane & 15, lane >> 4
                        pix = p0 + 16 * u + c
                        if pix < M:
                            v = np.maximum(np.concatenate([acc3[2 * s][u][lane], acc3[2 * s + 1][u][lane]]), 0)
                            aout[pix, 32 * s + 8 * q:32 * s + 8 * q + 8] = v
    # ---- reference
    a4 = ain.reshape(B, H, W, 64)
    pad = np.zeros((B, H + 2, W + 2, 64), np.float32)
    pad[:, 1:-1, 1:-1] = a4
    h = np.zeros((B, H, W, 64), np.float32)
    for t in range(9):
        h += np.einsum("bhwc,nc->bhwn", pad[:, t // 3:t // 3 + H, t % 3:t % 3 + W], w2b[:, t])
    h = np.maximum(h + b2b, 0).reshape(M, 64)
    xr = np.maximum(h @ w2c.T + b2c + xin, 0)
    ar = np.maximum(xr @ w2a.T + b2a, 0)
    err = float(np.abs(xout - xr).max())
    if tail:
        err = max(err, float(np.abs(aout - ar).max()))
    return err


if __name__ == "__main__":
    worst = 0.0
    for (B, H, W, tail, seed) in [(1, 5, 9, True, 0), (2, 4, 7, False, 1), (1, 3, 40, True, 2)]:
        e = run(B, H, W, tail, seed)
        print("B=%d H=%d W=%d tail=%s: max |error| %.1f" % (B, H, W, tail, e))
        worst = max(worst, e)
    raise SystemExit(1 if worst > 0 else 0)
