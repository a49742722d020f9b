"""CPU emulation of conv_gemm8_kernel's addressing, ring slots, staging cursors and issue schedule (csrc/rtn_conv_gemm8.hip), in the
style of tools/emulate_halo8.py: LDS as a byte array, every LDS-DMA piece queued at its issue point and landed either at once
("early") or only when a counted vmcnt wait of its wave retires it ("late").  Small-integer data, exact in float32, against a direct
evaluation of out = relu([x1(strided) | x2(strided)] . W^T + b).    python tools/emulate_gemm8.py  -> max |error| 0"""
import sys

import numpy as np

OOB = 0xFFFF0000
STAGE, B_BASE = 32768, 3 * 32768


def bf16_bytes(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16
    return u.astype(np.uint16).view(np.uint8)


def from_bf16(b):
    return (b.view(np.uint16).astype(np.uint32) << 16).view(np.float32)


def run(B, Ho, Wo, C1, step1, C2, step2, N, MI, grid, relu, late, seed):
    rng = np.random.RandomState(seed)
    R = 64 * MI
    H1, W1 = (Ho - 1) * step1 + 1, (Wo - 1) * step1 + 2          # a little wider than needed: strides are not the dense ones
    x1 = rng.randint(-3, 4, size=(B, H1, W1, C1)).astype(np.float32)
    dual = C2 > 0
    if dual:
        H2, W2 = (Ho - 1) * step2 + 2, (Wo - 1) * step2 + 1
        x2 = rng.randint(-3, 4, size=(B, H2, W2, C2)).astype(np.float32)
    K = C1 + C2
    w = rng.randint(-2, 3, size=(N, K)).astype(np.float32)
    bias = rng.randint(-4, 5, size=N).astype(np.float32)
    M = B * Ho * Wo
    Kbytes, nk, nk1 = K * 2, K * 2 // 128, C1 * 2 // 128
    ntn = N // 256
    ntm = -(-M // R)
    ntiles = ntm * ntn
    grid = min(grid, ntiles)
    x1b, wb = bf16_bytes(x1).reshape(-1), bf16_bytes(w).reshape(-1)
    x2b = bf16_bytes(x2).reshape(-1) if dual else None
    out = np.full((M, N), np.nan, np.float32)
    lanes = np.arange(64)
    lr, sc, lrow, kq = lanes >> 3, (lanes & 7) ^ (lanes >> 3), lanes & 15, lanes >> 4

    def load16(buf, off):
        if off >= OOB or off + 16 > len(buf):
            return np.zeros(16, np.uint8)
        return buf[off:off + 16]

    for bid in range(grid):
        lds = np.zeros(160 * 1024, np.uint8)
        queue = [[] for _ in range(8)]

        def dma(wave, buf, voff, soff, dst):
            data = np.stack([load16(buf, int(v) + soff) if v < OOB else np.zeros(16, np.uint8) for v in voff])
            if late:
                queue[wave].append((dst, data))
            else:
                lds[dst:dst + 1024] = data.reshape(-1)

        def vmcnt(wave, n):
            while len(queue[wave]) > n:
                dst, data = queue[wave].pop(0)
                lds[dst:dst + 1024] = data.reshape(-1)

        q_, r_, xcd = grid >> 3, grid & 7, bid & 7
        tile = (xcd * (q_ + 1) if xcd < r_ else r_ * (q_ + 1) + (xcd - r_) * q_) + (bid >> 3)
        st = {"hoff1": None, "hoff2": None, "wrow": None}

        def a_tile(T):
            h1 = np.full((8, MI, 64), OOB, np.int64)
            h2 = np.full((8, MI, 64), OOB, np.int64)
            mt = T // ntn
            for wave in range(8):
                for i in range(MI):
                    m = mt * R + (i * 8 + wave) * 8 + lr
                    ok = (T < ntiles) & (m < M)
                    mm = np.where(ok, m, 0)
                    b, rem = mm // (Ho * Wo), mm % (Ho * Wo)
                    oy, ox = rem // Wo, rem % Wo
                    o1 = ((b * H1 + oy * step1) * W1 + ox * step1) * C1 * 2 + sc * 16
                    h1[wave, i] = np.where(ok, o1, OOB)
                    if dual:
                        o2 = ((b * H2 + oy * step2) * W2 + ox * step2) * C2 * 2 + sc * 16
                        h2[wave, i] = np.where(ok, o2, OOB)
            st["hoff1"], st["hoff2"] = h1, h2

        def b_tile(T):
            nt = T % ntn if T < ntiles else 0
            wr = np.zeros((8, 4, 64), np.int64)
            for wave in range(8):
                for d in range(4):
                    P = d * 64 + wave * 8 + lr
                    nrow = nt * 256 + (P >> 7) * 128 + 8 * (P & 15) + ((P >> 4) & 7)
                    wr[wave, d] = nrow * Kbytes + sc * 16
            st["wrow"] = wr

        def stage_a(wave, i, k, slot):
            if dual and k >= nk1:
                dma(wave, x2b, st["hoff2"][wave, i], (k - nk1) * 128, slot + wave * 1024 + i * 8192)
            else:
                dma(wave, x1b, st["hoff1"][wave, i], k * 128, slot + wave * 1024 + i * 8192)

        def stage_b(wave, d, k, slot):
            dma(wave, wb, st["wrow"][wave, d], k * 128, slot + wave * 1024 + d * 8192)

        cur = {"ta": tile, "ka": 0, "tb": tile, "kb": 0, "a_st": 0, "b_st": 0}

        def adv_a():
            cur["a_st"] = 0 if cur["a_st"] == 2 * STAGE else cur["a_st"] + STAGE
            cur["ka"] += 1
            if cur["ka"] == nk:
                cur["ka"] = 0
                cur["ta"] += grid
                a_tile(cur["ta"])

        def adv_b():
            cur["b_st"] ^= STAGE
            cur["kb"] += 1
            if cur["kb"] == nk:
                cur["kb"] = 0
                cur["tb"] += grid
                b_tile(cur["tb"])

        a_tile(tile)
        b_tile(tile)
        for s in range(2):
            for wave in range(8):
                for i in range(MI):
                    stage_a(wave, i, cur["ka"], cur["a_st"])
            adv_a()
        for wave in range(8):
            for d in range(4):
                stage_b(wave, d, cur["kb"], B_BASE + cur["b_st"])
        adv_b()
        for wave in range(8):
            vmcnt(wave, 0)
        arow = np.zeros((8, MI, 64), np.int64)
        b_lane = np.zeros((8, 64), np.int64)
        for wave in range(8):
            wm, wn = wave >> 1, wave & 1
            for i in range(MI):
                row = wm * 16 * MI + i * 16 + lrow
                arow[wave, i] = row * 128 + ((kq ^ (row & 7)) << 4)
            b_lane[wave] = B_BASE + (wn * 128 + lrow) * 128 + ((kq ^ (lrow & 7)) << 4)

        def frag(addr):
            return np.stack([from_bf16(lds[a:a + 16].copy()) for a in addr])

        def mfma(acc, fa, fb):
            A = np.zeros((16, 32), np.float32)
            Bm = np.zeros((32, 16), np.float32)
            for l in range(64):
                A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = fa[l]
                Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = fb[l]
            D = A @ Bm
            for l in range(64):
                for r in range(4):
                    acc[l, r] += D[4 * (l >> 4) + r, l & 15]

        a_cur, b_cur = 0, 0
        while tile < ntiles:
            mt, nt = tile // ntn, tile % ntn
            m0, n0 = mt * R, nt * 256
            acc = np.zeros((8, MI, 8, 64, 4), np.float32)
            for wave in range(8):
                wn = wave & 1
                for j in range(8):
                    acc[wave, :, j] = bias[n0 + wn * 128 + 8 * lrow + j][None, :, None]
            for k in range(nk):
                fa = [[None] * MI for _ in range(8)]
                fb = [[None] * 4 for _ in range(8)]

                def lda(ks):
                    for wave in range(8):
                        for i in range(MI):
                            fa[wave][i] = frag(a_cur + (arow[wave, i] ^ (ks * 64)))

                def ldb(ks, half):
                    for wave in range(8):
                        for j in range(4):
                            fb[wave][j] = frag(b_cur + (b_lane[wave] ^ (ks * 64)) + (half * 4 + j) * 2048)

                def mm(half):
                    for wave in range(8):
                        for j in range(4):
                            for i in range(MI):
                                mfma(acc[wave, i, half * 4 + j], fa[wave][i], fb[wave][j])

                lda(0); ldb(0, 0); mm(0)
                ldb(0, 1)
                for wave in range(8):
                    stage_b(wave, 0, cur["kb"], B_BASE + cur["b_st"]); stage_b(wave, 1, cur["kb"], B_BASE + cur["b_st"])
                mm(1)
                lda(1); ldb(1, 0)
                for wave in range(8):
                    stage_b(wave, 2, cur["kb"], B_BASE + cur["b_st"]); stage_b(wave, 3, cur["kb"], B_BASE + cur["b_st"])
                adv_b()
                for wave in range(8):
                    stage_a(wave, 0, cur["ka"], cur["a_st"])
                mm(0)
                ldb(1, 1)
                for wave in range(8):
                    for i in range(1, MI):
                        stage_a(wave, i, cur["ka"], cur["a_st"])
                adv_a()
                for wave in range(8):
                    vmcnt(wave, MI)
                mm(1)
                a_cur = 0 if a_cur == 2 * STAGE else a_cur + STAGE
                b_cur ^= STAGE
            for wave in range(8):
                wm, wn = wave >> 1, wave & 1
                for i in range(MI):
                    for r in range(4):
                        for l in range(64):
                            q, c = l >> 4, l & 15
                            m = m0 + wm * 16 * MI + i * 16 + q * 4 + r
                            ncol = n0 + wn * 128 + 8 * c
                            if ncol < N and m < M:
                                v = acc[wave, i, :, l, r].copy()
                                out[m, ncol:ncol + 8] = np.maximum(v, 0) if relu else v
            tile += grid
        for wave in range(8):
            vmcnt(wave, 0)
    # reference
    xs = x1[:, :(Ho - 1) * step1 + 1:step1, :(Wo - 1) * step1 + 1:step1].reshape(M, C1)
    if dual:
        xs = np.concatenate([xs, x2[:, :(Ho - 1) * step2 + 1:step2, :(Wo - 1) * step2 + 1:step2].reshape(M, C2)], axis=1)
    ref = xs @ w.T + bias
    if relu:
        ref = np.maximum(ref, 0)
    assert not np.isnan(out).any(), "unwritten outputs"
    return float(np.abs(out - ref).max())


if __name__ == "__main__":
    worst = 0.0
    cases = [  # B, Ho, Wo, C1, step1, C2, step2, N, MI, grid, relu
        (1, 9, 23, 128, 1, 0, 1, 256, 3, 1, True),       # 2 tiles on one workgroup, 2 K steps each
        (2, 5, 7, 64, 2, 0, 1, 512, 2, 3, False),        # stride-2 sampling, two N tiles, 3 workgroups
        (1, 6, 11, 64, 1, 128, 2, 256, 3, 2, True),      # dual source, the second one strided
        (1, 4, 5, 192, 1, 0, 1, 256, 2, 1, True),        # 3 K steps: the A ring wraps inside a tile
    ]
    for cs in cases:
        for late in (False, True):
            e = run(*cs, late=late, seed=1)
            worst = max(worst, e)
        print(cs, "max |error| %.1f" % e)
    sys.exit(1 if worst > 0 else 0)
