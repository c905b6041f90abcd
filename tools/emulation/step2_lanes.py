"""Host-only (numpy).  Lane-level emulation of the MFMA call sites of k_big_step2 (role B') and step2_border (role C'): every operand index expression
of the kernel transcribed as it stands, v_mfma_f64_16x16x4_f64 emulated with the operand layout the running kernels of mf_numeric.hip rely on
    A[i = l & 15][kk = l >> 4],  B[kk = l >> 4][j = l & 15],  D[row = (l >> 4) + 4 reg][col = l & 15],
one workgroup of role B' (pivot work + 96 rows) and the workgroups of role C' run on random data and compared with the matrix expressions they stand for.
What this checks is the lane / register bookkeeping (which the matrix-level emulations cannot see); the pivot inverse itself is taken from numpy."""
import numpy as np
NB, LDP, LDX, LD2, LDT, MT2, KW = 32, 33, 33, 65, 17, 2, 64
rng = np.random.default_rng(7)
LO = np.arange(64) & 15
HI = np.arange(64) >> 4
def mfma(a, b, acc):
    """a, b: per-lane operands (64,), acc: (64, 4) accumulators"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[LO[l], HI[l]] = a[l]; B[HI[l], LO[l]] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for r in range(4): out[l, r] += D[HI[l] + 4 * r, LO[l]]
    return out
def lanes(f):
    return np.array([f(LO[l], HI[l]) for l in range(64)])

def role_b(N, nc, kb, r0, wgid=0):
    A0 = rng.normal(size=(N, N)); A0 = A0 @ A0.T + N * np.eye(N)
    F = np.tril(A0) + np.triu(rng.normal(size=(N, N)), 1)          # garbage above the diagonal
    F[:, :kb + 64 if kb >= 0 else 0] = np.tril(rng.normal(size=(N, N)))[:, :kb + 64 if kb >= 0 else 0] * 0.1 + F[:, :kb + 64 if kb >= 0 else 0] * 0   # some "final" P columns
    w = min(KW, nc - kb) if kb >= 0 else 0
    kb1 = kb + w if kb >= 0 else 0
    wq = min(KW, nc - kb1) if kb1 < nc else 0
    w1 = min(NB, wq); w2 = wq - w1
    Fc = lambda r, c: F[r, c]
    Lp = np.zeros(KW * LD2); Aq = np.zeros(KW * LD2)
    for e in range(KW * KW):
        k, q = e >> 6, e & 63
        vl = Fc(min(kb1 + q, N - 1), max(kb, 0) + min(k, max(w, 1) - 1)); vd = Fc(min(kb1 + q, N - 1), kb1 + min(k, wq - 1))
        Lp[k * LD2 + q] = vl if (k < w and q < wq) else 0.0
        Aq[k * LD2 + q] = vd if (k < wq and q < wq and q >= k) else 0.0
    # expectation at matrix level
    LpM = np.array([[Lp[k * LD2 + q] for q in range(64)] for k in range(64)])           # [k][q]
    AqM0 = np.array([[Aq[c * LD2 + q] for c in range(64)] for q in range(64)])          # (q, c), lower
    AqE = AqM0 - np.tril(LpM.T @ LpM) * (1 if w > 0 else 0)
    def pairUpdateTile(ti, tj):
        acc = np.zeros((64, 4))
        for ks in range(KW // 4):
            acc = mfma(lanes(lambda lo, hi: Lp[(4 * ks + hi) * LD2 + 16 * ti + lo]), lanes(lambda lo, hi: Lp[(4 * ks + hi) * LD2 + 16 * tj + lo]), acc)
        for l in range(64):
            c = 16 * tj + LO[l]
            for r in range(4):
                q = 16 * ti + HI[l] + 4 * r
                if q >= c: Aq[c * LD2 + q] -= acc[l, r]
    if w > 0:
        for (ti, tj) in [(0, 0), (1, 0), (1, 1)]: pairUpdateTile(ti, tj)
        if w2 > 0:
            for (ti, tj) in [(2, 0), (3, 1), (3, 3), (2, 1), (2, 2), (3, 0), (3, 2)]: pairUpdateTile(ti, tj)
    AqM = np.array([[Aq[c * LD2 + q] for c in range(64)] for q in range(64)])
    mask = np.tril(np.ones((64, 64))) if w2 > 0 else np.pad(np.tril(np.ones((32, 32))), ((0, 32), (0, 32)))
    e1 = np.abs((AqM - AqE) * mask).max()
    # pivot inverses from numpy (identity-padded), in the kernel's Xs layout
    def inv_block(M, wd):
        B = np.eye(32); B[:wd, :wd] = M[:wd, :wd]; B = np.tril(B) + np.tril(B, -1).T
        return np.linalg.inv(np.linalg.cholesky(B))
    X1 = inv_block(AqM[:32, :32], w1)
    Xs1 = np.zeros(NB * LDX)
    for c in range(32):
        for r in range(32): Xs1[c * LDX + r] = X1[r, c]
    # rows of one workgroup: wave wv, tile mt
    Rb = kb1 + wq
    errs = [e1]
    L21s = np.zeros(NB * LDP)
    if w2 > 0:
        for tw in range(4):
            tc, tq = tw >> 1, tw & 1
            o = np.zeros((64, 4))
            for ks in range(NB // 4):
                o = mfma(lanes(lambda lo, hi: Xs1[(4 * ks + hi) * LDX + 16 * tc + lo]), lanes(lambda lo, hi: Aq[(4 * ks + hi) * LD2 + NB + 16 * tq + lo]), o)
            for l in range(64):
                q = 16 * tq + LO[l]
                for i in range(4):
                    c = 16 * tc + HI[l] + 4 * i
                    L21s[c * LDP + q] = o[l, i]
        L21 = np.array([[L21s[c * LDP + q] for c in range(32)] for q in range(32)])
        errs.append(np.abs(L21 - AqM[32:, :32] @ X1.T).max())
        for tw in range(3):
            ti, tj = int(tw >= 1), int(tw == 2)
            acc = np.zeros((64, 4))
            for ks in range(NB // 4):
                acc = mfma(lanes(lambda lo, hi: L21s[(4 * ks + hi) * LDP + 16 * ti + lo]), lanes(lambda lo, hi: L21s[(4 * ks + hi) * LDP + 16 * tj + lo]), acc)
            for l in range(64):
                c = 16 * tj + LO[l]
                for r in range(4):
                    q = 16 * ti + HI[l] + 4 * r
                    if q >= c: Aq[(NB + c) * LD2 + NB + q] -= acc[l, r]
        A22 = np.array([[Aq[(NB + c) * LD2 + NB + q] for c in range(32)] for q in range(32)])
        errs.append(np.abs(np.tril(A22 - (AqM[32:, 32:] - L21 @ L21.T))).max())
        X2 = inv_block(A22, w2)
        Xs2 = np.zeros(NB * LDX)
        for c in range(32):
            for r in range(32): Xs2[c * LDX + r] = X2[r, c]
    for wv in range(3):
        Rw = Rb + r0 + 16 * MT2 * wv
        if Rw >= N: continue
        for mt in range(MT2):
            rowl = lambda lo: min(Rw + 16 * mt + lo, N - 1)
            dt = [np.zeros((64, 4)) for _ in range(4)]
            for ct in range(4):
                for l in range(64):
                    for i in range(4):
                        c = 16 * ct + HI[l] + 4 * i
                        v = Fc(rowl(LO[l]), kb1 + min(c, max(wq, 1) - 1))
                        dt[ct][l, i] = v if c < wq else 0.0
            pv = [lanes(lambda lo, hi: Fc(rowl(lo), min(max(kb, 0) + 4 * ks + hi, N - 1))) for ks in range(KW // 4)]
            if w > 0:
                for ks in range(KW // 4):
                    for ct in range(4):
                        dt[ct] = mfma(lanes(lambda lo, hi: -Lp[(4 * ks + hi) * LD2 + 16 * ct + lo]), pv[ks], dt[ct])
            rows = [Rw + 16 * mt + lo for lo in range(16)]
            valid = [m for m in rows if m < N]
            if not valid: continue   # (a tile past the last row: computed on clamped addresses, stored nowhere)
            rawM = np.array([[F[min(m, N - 1), kb1 + c] if c < wq else 0.0 for c in range(64)] for m in rows])
            PM = np.array([[F[min(m, N - 1), max(kb, 0) + k] for k in range(64)] for m in rows]) if w > 0 else np.zeros((16, 64))
            DE = rawM - PM @ LpM                                     # (m, c)
            Dm = np.zeros((16, 64))
            for ct in range(4):
                for l in range(64):
                    for i in range(4): Dm[LO[l], 16 * ct + HI[l] + 4 * i] = dt[ct][l, i]
            errs.append(np.abs(Dm - DE)[:len(valid)].max())
            a0 = np.zeros((64, 4)); a1 = np.zeros((64, 4))
            for ks in range(4):
                a0 = mfma(lanes(lambda lo, hi: Xs1[(4 * ks + hi) * LDX + lo]), dt[0][:, ks], a0)
                a1 = mfma(lanes(lambda lo, hi: Xs1[(4 * ks + hi) * LDX + 16 + lo]), dt[0][:, ks], a1)
            for ks in range(4):
                a1 = mfma(lanes(lambda lo, hi: Xs1[(16 + 4 * ks + hi) * LDX + 16 + lo]), dt[1][:, ks], a1)
            LQ1 = np.zeros((16, 32))
            for l in range(64):
                for i in range(4):
                    LQ1[LO[l], HI[l] + 4 * i] = a0[l, i]; LQ1[LO[l], 16 + HI[l] + 4 * i] = a1[l, i]
            errs.append(np.abs(LQ1 - DE[:, :32] @ X1.T)[:len(valid)].max())
            if w2 > 0:
                for ks in range(4):
                    for ct in range(2):
                        dt[2 + ct] = mfma(lanes(lambda lo, hi: -L21s[(4 * ks + hi) * LDP + 16 * ct + lo]), a0[:, ks], dt[2 + ct])
                        dt[2 + ct] = mfma(lanes(lambda lo, hi: -L21s[(16 + 4 * ks + hi) * LDP + 16 * ct + lo]), a1[:, ks], dt[2 + ct])
                D2 = np.zeros((16, 32))
                for ct in range(2):
                    for l in range(64):
                        for i in range(4): D2[LO[l], 16 * ct + HI[l] + 4 * i] = dt[2 + ct][l, i]
                D2E = DE[:, 32:] - (DE[:, :32] @ X1.T) @ L21.T
                errs.append(np.abs(D2 - D2E)[:len(valid)].max())
                b0 = np.zeros((64, 4)); b1 = np.zeros((64, 4))
                for ks in range(4):
                    b0 = mfma(lanes(lambda lo, hi: Xs2[(4 * ks + hi) * LDX + lo]), dt[2][:, ks], b0)
                    b1 = mfma(lanes(lambda lo, hi: Xs2[(4 * ks + hi) * LDX + 16 + lo]), dt[2][:, ks], b1)
                for ks in range(4):
                    b1 = mfma(lanes(lambda lo, hi: Xs2[(16 + 4 * ks + hi) * LDX + 16 + lo]), dt[3][:, ks], b1)
                LQ2 = np.zeros((16, 32))
                for l in range(64):
                    for i in range(4):
                        LQ2[LO[l], HI[l] + 4 * i] = b0[l, i]; LQ2[LO[l], 16 + HI[l] + 4 * i] = b1[l, i]
                errs.append(np.abs(LQ2 - D2E @ X2.T)[:len(valid)].max())
    return max(errs)

def role_c(N, nc, kb, c0):
    """one 16-column tile workgroup of step2_border (c0 < kb) and the pair's own block (c0 == kb), against the matrix expressions"""
    w = min(2 * NB, nc - kb); wa = min(NB, w); wb = w - wa
    F = rng.normal(size=(N, N))                        # L values where they are read; the stash of L21 above the diagonal
    Xm = np.tril(rng.normal(size=(nc, nc)))            # X rows < kb final
    XT = Xm.T.copy()
    X1 = np.tril(rng.normal(size=(32, 32))); X2 = np.tril(rng.normal(size=(32, 32)))
    blk1 = np.zeros(1024); blk2 = np.zeros(1024)
    for c in range(32):
        for r in range(32): blk1[c * 32 + r] = X1[r, c]; blk2[c * 32 + r] = X2[r, c]
    L21 = np.array([[F[kb + n, min(kb + NB + q, N - 1)] for n in range(32)] for q in range(32)])   # from the stash (transposed)
    if c0 == kb:
        Xa = np.zeros(NB * LDX); Xb = np.zeros(NB * LDX); Us = np.zeros(NB * LDP)
        for e in range(1024):
            c, r = e >> 5, e & 31
            Xa[c * LDX + r] = blk1[e]; Xb[c * LDX + r] = blk2[e] if wb > 0 else 0.0
        if wb == 0: return 0.0
        out = np.zeros((32, 32))
        for wv in range(4):
            tc, tq = wv >> 1, wv & 1
            l21 = [lanes(lambda lo, hi: F[kb + 4 * ks + hi, min(kb + NB + 16 * tq + lo, N - 1)]) for ks in range(8)]
            u = np.zeros((64, 4))
            for ks in range(8):
                u = mfma(lanes(lambda lo, hi: Xa[(16 * tc + lo) * LDX + 4 * ks + hi]), np.where(16 * tq + LO < wb, l21[ks], 0.0), u)
            for l in range(64):
                for i in range(4): Us[(16 * tq + LO[l]) * LDP + 16 * tc + HI[l] + 4 * i] = u[l, i]
        for wv in range(4):
            tc, tq = wv >> 1, wv & 1
            o = np.zeros((64, 4))
            for ks in range(8):
                o = mfma(lanes(lambda lo, hi: Us[(4 * ks + hi) * LDP + 16 * tc + lo]), lanes(lambda lo, hi: Xb[(4 * ks + hi) * LDX + 16 * tq + lo]), o)
            for l in range(64):
                r2 = 16 * tq + LO[l]
                if r2 < wb:
                    for i in range(4): out[r2, 16 * tc + HI[l] + 4 * i] = -o[l, i]
        L21m = L21.copy(); L21m[wb:, :] = 0.0
        E = -X2 @ (L21m @ X1)
        return np.abs(out[:wb] - E[:wb]).max()
    # tile workgroup
    red = np.zeros((4, 4, 256)); Ts = np.zeros(2 * NB * LDT); XR1s = np.zeros(NB * LDT); Us = np.zeros(NB * LDT)
    cc = lambda lo: c0 + lo
    rr = lambda b, lo: kb + 16 * b + lo
    for wv in range(4):
        acc = [np.zeros((64, 4)) for _ in range(4)]
        k0 = c0 + 16 * wv
        while k0 < kb:
            for kk in (k0, k0 + 64, k0 + 128):
                if kk < kb:
                    for ks in range(4):
                        kf = lambda hi: kk + 4 * ks + hi
                        ra = lanes(lambda lo, hi: XT[cc(lo), min(kf(hi), nc - 1)])
                        ma = lanes(lambda lo, hi: 1.0 if (kf(hi) < kb and kf(hi) >= cc(lo)) else 0.0) * ra
                        for b in range(4):
                            rb = lanes(lambda lo, hi: F[min(rr(b, lo), N - 1), min(kf(hi), nc - 1)])
                            mb = lanes(lambda lo, hi: 1.0 if (kf(hi) < kb and rr(b, lo) < kb + w) else 0.0) * rb
                            acc[b] = mfma(ma, mb, acc[b])
            k0 += 192
        for b in range(4):
            for i in range(4):
                for l in range(64): red[wv, b, 64 * i + l] = acc[b][l, i]
    for wv in range(4):
        for l in range(64):
            for i in range(4):
                Ts[(16 * wv + LO[l]) * LDT + HI[l] + 4 * i] = ((red[0, wv, 64 * i + l] + red[1, wv, 64 * i + l]) + red[2, wv, 64 * i + l]) + red[3, wv, 64 * i + l]
    T = np.array([[Ts[r * LDT + c] for c in range(16)] for r in range(64)])
    Lrows = np.array([[F[min(kb + r, N - 1), k] if r < w else 0.0 for k in range(kb)] for r in range(64)])
    TE = Lrows[:, c0:] @ np.tril(Xm[:kb, :kb])[c0:, c0:c0 + 16]
    e = [np.abs(T - TE).max()]
    XR1 = np.zeros((32, 16)); XR2 = np.zeros((32, 16))
    for wv in range(2):
        bq = wv & 1
        xd = [lanes(lambda lo, hi: blk1[(4 * ks + hi) * NB + 16 * bq + lo]) for ks in range(8)]
        o = np.zeros((64, 4))
        for ks in range(8):
            o = mfma(lanes(lambda lo, hi: Ts[(4 * ks + hi) * LDT + lo]), np.where(4 * ks + HI <= 16 * bq + LO, xd[ks], 0.0), o)
        for l in range(64):
            r = 16 * bq + LO[l]
            for i in range(4):
                c = HI[l] + 4 * i
                XR1s[r * LDT + c] = -o[l, i] if r < wa else 0.0
                if r < wa: XR1[r, c] = -o[l, i]
    E1 = -X1 @ TE[:32]
    e.append(np.abs(XR1[:wa] - E1[:wa]).max())
    if wb > 0:
        for wv in (2, 3):
            bq = wv & 1
            l21 = [lanes(lambda lo, hi: F[kb + 4 * ks + hi, min(kb + NB + 16 * bq + lo, N - 1)]) for ks in range(8)]
            u = np.zeros((64, 4))
            for ks in range(8):
                u = mfma(lanes(lambda lo, hi: XR1s[(4 * ks + hi) * LDT + lo]), np.where(16 * bq + LO < wb, l21[ks], 0.0), u)
            for l in range(64):
                k2 = 16 * bq + LO[l]
                for i in range(4): Us[k2 * LDT + HI[l] + 4 * i] = u[l, i] + Ts[(NB + k2) * LDT + HI[l] + 4 * i]
        for wv in (2, 3):
            bq = wv & 1
            xd = [lanes(lambda lo, hi: blk2[(4 * ks + hi) * NB + 16 * bq + lo]) for ks in range(8)]
            o = np.zeros((64, 4))
            for ks in range(8):
                o = mfma(lanes(lambda lo, hi: Us[(4 * ks + hi) * LDT + lo]), np.where(4 * ks + HI <= 16 * bq + LO, xd[ks], 0.0), o)
            for l in range(64):
                r2 = 16 * bq + LO[l]
                if r2 < wb:
                    for i in range(4): XR2[r2, HI[l] + 4 * i] = -o[l, i]
        L21m = L21.copy(); L21m[wb:, :] = 0.0
        XR1f = np.zeros((32, 16)); XR1f[:wa] = E1[:wa]
        E2 = -X2 @ (TE[32:] + L21m @ XR1f)
        e.append(np.abs(XR2[:wb] - E2[:wb]).max())
    return max(e)

if __name__ == "__main__":
    for (N, nc, kb, r0) in [(300, 256, 64, 0), (300, 256, 64, 96), (200, 128, -1, 0), (200, 161, 64, 0), (140, 129, 64, 0), (100, 97, 0, 0), (64, 40, -1, 0)]:
        print("role B'  N %d nc %d kb %d r0 %d: max deviation from the matrix expressions %.1e" % (N, nc, kb, r0, role_b(N, nc, kb, r0)))
    for (N, nc, kb, c0) in [(300, 256, 128, 0), (300, 256, 128, 112), (300, 256, 128, 128), (300, 225, 192, 48), (300, 225, 192, 192), (300, 200, 192, 16), (300, 200, 192, 192)]:
        print("role C'  N %d nc %d kb %d c0 %d: max deviation %.1e" % (N, nc, kb, c0, role_c(N, nc, kb, c0)))
