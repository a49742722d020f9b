"""CPU emulation of conv_halo8_kernel's ADDRESSING AND ISSUE SCHEDULE (csrc/rtn_conv_halo8.hip), transliterated statement by
statement: LDS as a byte array, LDS-DMA pieces as 64 x 16-byte copies with the source-side swizzle, fragment reads through the
swizzled addresses, the MFMA's lane -> element maps, the permuted weight rows and the register epilogue.  Checked against a
direct convolution on small integer data (exact in float32).

Every LDS-DMA piece is queued at its issue point and lands either immediately ("early") or only when a counted vmcnt wait of
ITS wave retires it ("late"): the two extremes of what the hardware may do.  A wrong buffer slot, a piece issued before the
last read of the buffer it overwrites, or a read ahead of the wait that covers it gives a wrong output in one of the two modes.
(The stagger between the two wave groups is not modelled: waves run phase by phase in lockstep.)

  python tools/emulate_halo8.py            # runs the cases below, prints max |error| per case (must be 0)
"""
import sys

import numpy as np

KW = 3
TM = 256 - KW
OOB = 0xFFFFFF00
B_STAGE, A_BASE, A_TOGGLE, ZERO_ROW = 32768, 3 * 32768, 0x18000 ^ 0x20000, 255 * 128


def bf16_bytes(a):
    """float array holding bf16-representable values -> uint8 view of the bf16 encoding"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16
    return u.astype(np.uint16).view(np.uint8)


def from_bf16(b):
    return (b.view(np.uint16).astype(np.uint32) << 16).view(np.float32)


class Emu:
    def __init__(self, groups, w, bias, N, KH, Cin, pad_t, pad_l, relu, grid, late):
        """groups: list of (x[B,H,W,Cin] float32 of bf16 values); w[N rows (256)][KH*KW*Cin]"""
        self.late = late
        self.groups = groups
        self.KH, self.nchunk, self.pad_t, self.pad_l, self.relu = KH, Cin * 2 // 128, pad_t, pad_l, relu
        self.Kbytes = KH * KW * Cin * 2
        self.pix_b = Cin * 2
        self.N = N
        self.wbytes = bf16_bytes(w).reshape(-1)
        self.bias = bias
        self.g = []
        tiles = 0
        for x in groups:
            B, H, W, _ = x.shape
            M = B * H * W
            self.g.append(dict(inb=bf16_bytes(x).reshape(-1), Hin=H, Win=W, M=M, tile_begin=tiles, row_b=W * Cin * 2,
                               out=np.full((M, N), np.nan, np.float32)))
            tiles += -(-M // TM)
        self.ntiles = tiles
        self.grid = min(grid, tiles)

    # ---- device memory
    def load16(self, buf, off):
        if off + 16 > len(buf) or off >= OOB:
            return np.zeros(16, np.uint8)
        return buf[off:off + 16]

    def run(self):
        for wg in range(self.grid):
            self.run_wg(wg)
        return [g["out"] for g in self.g]

    def group_of(self, T):
        gi = 0
        for i in range(1, len(self.g)):
            if T >= self.g[i]["tile_begin"]:
                gi = i
        return gi

    def run_wg(self, wg):
        lds = np.zeros(160 * 1024, np.uint8)
        lanes = np.arange(64)
        lr, sc = lanes >> 3, (lanes & 7) ^ (lanes >> 3)
        lrow, kq = lanes & 15, lanes >> 4
        G = self.KH * self.nchunk
        nchunk = self.nchunk
        queue = [[] for _ in range(8)]                 # per wave: pending pieces (dst, data[64,16])

        def dma(wave, buf, voff, soff, dst):
            data = np.stack([self.load16(buf, int(v) + soff) if v < OOB else np.zeros(16, np.uint8) for v in voff])
            if self.late:
                queue[wave].append((dst, data))
            else:
                lds[dst:dst + 1024] = data.reshape(-1)

        def vmcnt(wave, n):
            while len(queue[wave]) > n:
                dst, data = queue[wave].pop(0)
                lds[dst:dst + 1024] = data.reshape(-1)

        wrow_off = np.zeros((8, 4, 64), np.int64)
        for wave in range(8):
            for d in range(4):
                P = d * 64 + wave * 8 + lr
                nrow = (P >> 7) * 128 + 8 * (P & 15) + ((P >> 4) & 7)
                wrow_off[wave, d] = nrow * self.Kbytes + sc * 16
        st = {}

        def stage_tile(T):
            st["hiy"] = np.full((8, 4, 64), -(1 << 28), np.int64)
            st["hbase"] = np.zeros((8, 4, 64), np.int64)
            if T >= self.ntiles:
                st["g"] = None
                return
            gi = self.group_of(T)
            Gs = self.g[gi]
            st["g"] = Gs
            m0 = (T - Gs["tile_begin"]) * TM
            cells = Gs["Hin"] * Gs["Win"]
            for wave in range(8):
                for i in range(4):
                    h = (i * 8 + wave) * 8 + lr
                    f = m0 + h - self.pad_l
                    ok = (h < 255) & (f >= 0) & (f < Gs["M"])
                    rem = np.where(ok, f, 0) % cells
                    st["hiy"][wave, i] = np.where(ok, rem // Gs["Win"], -(1 << 28))
                    st["hbase"][wave, i] = np.where(ok, f * self.pix_b + sc * 16, 0)

        def stage_a(wave, i, kh, cc, abuf):
            Gs = st["g"]
            dy = kh - self.pad_t
            if Gs is None:
                voff = np.full(64, OOB, np.int64)
                buf = self.wbytes
            else:
                delta = dy * Gs["row_b"] + cc * 128
                y = st["hiy"][wave, i] + dy
                ok = (y >= 0) & (y < Gs["Hin"])
                voff = np.where(ok, st["hbase"][wave, i] + delta, OOB)
                buf = Gs["inb"]
            dma(wave, buf, voff, 0, abuf + wave * 1024 + i * 8192)

        def stage_b(wave, d, kcol, bst):
            dma(wave, self.wbytes, wrow_off[wave, d], kcol, bst * B_STAGE + wave * 1024 + d * 8192)

        def frag(addr):                                # [64 lanes] byte addresses -> [64, 8] bf16 values
            return np.stack([from_bf16(lds[a:a + 16].copy()) for a in addr])

        def mfma(acc, fa, fb):
            # lane l: A[row l&15][k = 8 (l>>4) + e], B[k = 8 (l>>4) + e][col l&15]; D lane (q, c): rows 4q + r, col c
            A = np.zeros((16, 32), np.float32)
            Bm = np.zeros((32, 16), np.float32)
            for l in range(64):
                A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = fa[l]
                Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = fb[l]
            D = A @ Bm
            for l in range(64):
                for r in range(4):
                    acc[l, r] += D[4 * (l >> 4) + r, l & 15]

        b_lane = np.zeros((8, 64), np.int64)
        for wave in range(8):
            wn = wave & 1
            b_lane[wave] = (wn * 128 + lrow) * 128 + ((kq ^ (lrow & 7)) << 4)

        tile = wg
        stage_tile(tile)
        for wave in range(8):
            for i in range(4):
                stage_a(wave, i, 0, 0, A_BASE)
            for d in range(4):
                stage_b(wave, d, 0, 0)
            for d in range(3):
                stage_b(wave, d, nchunk * 128, 1)
            vmcnt(wave, 3)
        a_cur = A_BASE
        kw_stride = nchunk * 128
        while tile < self.ntiles:
            gi = self.group_of(tile)
            Gc = self.g[gi]
            m0 = (tile - Gc["tile_begin"]) * TM
            cells = Gc["Hin"] * Gc["Win"]
            arow = np.zeros((8, KW, 4, 64), np.int64)
            for wave in range(8):
                wm = wave >> 1
                for i in range(4):
                    rloc = wm * 64 + i * 16 + lrow
                    m = np.minimum(m0 + rloc, Gc["M"] - 1)
                    x = (m % cells) % Gc["Win"]
                    for kw in range(KW):
                        rr = rloc + kw
                        xx = x + kw - self.pad_l
                        off = (rloc >= TM) | (xx < 0) | (xx >= Gc["Win"])
                        arow[wave, kw, i] = a_cur + np.where(off, ZERO_ROW, rr * 128 + ((kq ^ (rr & 7)) << 4))
            acc = np.zeros((8, 4, 8, 64, 4), np.float32)
            for wave in range(8):
                wn = wave & 1
                for j in range(8):
                    acc[wave, :, j] = self.bias[wn * 128 + 8 * lrow + j][None, :, None]
            kh, cc = 0, 0
            for g in range(G):
                kh1, cc1 = kh, cc + 1
                if cc1 == nchunk:
                    cc1, kh1 = 0, kh + 1
                if g + 1 == G:
                    kh1 = cc1 = 0
                    stage_tile(tile + self.grid)
                kcol_g = (kh * KW * nchunk + cc) * 128
                kcol_g1 = (kh1 * KW * nchunk + cc1) * 128
                for kwi in range(KW):
                    kc_n1 = kcol_g + (kwi + 1) * kw_stride if kwi + 1 < KW else kcol_g1
                    kc_n2 = kcol_g + (kwi + 2) * kw_stride if kwi + 2 < KW else kcol_g1 + (kwi + 2 - KW) * kw_stride
                    fa = [[None] * 4 for _ in range(8)]
                    fb = [[None] * 4 for _ in range(8)]

                    def lda(ks):
                        for wave in range(8):
                            for i in range(4):
                                fa[wave][i] = frag(arow[wave, kwi, i] ^ (ks * 64))

                    def ldb(ks, half):
                        for wave in range(8):
                            for j in range(4):
                                fb[wave][j] = frag((b_lane[wave] ^ (ks * 64)) + kwi * B_STAGE + (half * 4 + j) * 2048)

                    def mm(half):
                        for wave in range(8):
                            for j in range(4):
                                for i in range(4):
                                    mfma(acc[wave, i, half * 4 + j], fa[wave][i], fb[wave][j])

                    # phase 1
                    lda(0); ldb(0, 0)
                    for wave in range(8):
                        stage_b(wave, 3, kc_n1, (kwi + 1) % 3)
                    mm(0)
                    # phase 2
                    ldb(0, 1)
                    for wave in range(8):
                        stage_b(wave, 0, kc_n2, (kwi + 2) % 3)
                        if kwi < 2:
                            stage_a(wave, 2 * kwi, kh1, cc1, a_cur ^ A_TOGGLE)
                    mm(1)
                    # phase 3
                    lda(1); ldb(1, 0)
                    for wave in range(8):
                        stage_b(wave, 1, kc_n2, (kwi + 2) % 3)
                    mm(0)
                    # phase 4
                    ldb(1, 1)
                    for wave in range(8):
                        stage_b(wave, 2, kc_n2, (kwi + 2) % 3)
                        if kwi < 2:
                            stage_a(wave, 2 * kwi + 1, kh1, cc1, a_cur ^ A_TOGGLE)
                        vmcnt(wave, 5 if kwi < 2 else 3)
                    mm(1)
                a_cur ^= A_TOGGLE
                arow = np.where(True, arow ^ A_TOGGLE, arow)
                kh, cc = kh1, cc1
            # epilogue
            for wave in range(8):
                wm, wn = wave >> 1, wave & 1
                for i in range(4):
                    for r in range(4):
                        for l in range(64):
                            q, c = l >> 4, l & 15
                            rloc = wm * 64 + i * 16 + q * 4 + r
                            m = m0 + rloc
                            ncol = wn * 128 + 8 * c
                            if ncol < self.N and rloc < TM and m < Gc["M"]:
                                v = acc[wave, i, :, l, r].copy()
                                if self.relu:
                                    v = np.maximum(v, 0)
                                Gc["out"][m, ncol:ncol + 8] = v
            tile += self.grid
        for wave in range(8):
            vmcnt(wave, 0)


def reference(x, w, bias, N, KH, Cin, pad_t, pad_l, relu):
    B, H, W, _ = x.shape
    wk = w[:N].reshape(N, KH, KW, Cin)
    xp = np.zeros((B, H + KH, W + KW, Cin), np.float32)
    xp[:, pad_t:pad_t + H, pad_l:pad_l + W] = x
    out = np.zeros((B, H, W, N), np.float32)
    for kh in range(KH):
        for kw in range(KW):
            out += np.einsum("bhwc,nc->bhwn", xp[:, kh:kh + H, kw:kw + W], wk[:, kh, kw])
    out += bias[:N]
    if relu:
        out = np.maximum(out, 0)
    return out.reshape(-1, N)


def case(name, shapes, Cin, N, KH, pad_t, pad_l, relu, grid, seed):
    rng = np.random.RandomState(seed)
    xs = [rng.randint(-3, 4, size=s + (Cin,)).astype(np.float32) for s in shapes]
    w = np.zeros((256, KH * KW * Cin), np.float32)
    w[:N] = rng.randint(-2, 3, size=(N, KH * KW * Cin))
    bias = np.zeros(256, np.float32)
    bias[:N] = rng.randint(-5, 6, size=N)
    worst = 0.0
    for late in (False, True):
        outs = Emu(xs, w, bias, N, KH, Cin, pad_t, pad_l, relu, grid, late).run()
        for x, o in zip(xs, outs):
            ref = reference(x, w, bias, N, KH, Cin, pad_t, pad_l, relu)
            assert not np.isnan(o).any(), "%s: unwritten outputs (late=%s)" % (name, late)
            worst = max(worst, float(np.abs(o - ref).max()))
    print("%-40s max |error| %.1f" % (name, worst))
    return worst


if __name__ == "__main__":
    bad = 0
    bad += case("1 group, 2 tiles on 1 workgroup", [(1, 20, 23)], 64, 256, 3, 1, 1, True, 1, 0) > 0
    bad += case("3 levels, 128 ch (2 chunks), 2 wgs", [(2, 9, 14), (2, 5, 7), (1, 3, 4)], 128, 136, 3, 1, 1, False, 2, 1) > 0
    bad += case("KH=1 row conv, pad_l=0", [(1, 16, 19)], 64, 200, 1, 0, 0, True, 3, 2) > 0
    sys.exit(1 if bad else 0)
