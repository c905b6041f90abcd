"""Host-only (numpy).  Tile-level emulation of the two-panels-per-launch step (roles A', B', C') on a dense front: checks the algebra and the launch invariants."""
import numpy as np
rng = np.random.default_rng(0)
def spd(n):
    A = rng.normal(size=(n, n)); return A @ A.T + n * np.eye(n)
def run(N, nc):
    A = spd(N)
    F = np.tril(A).copy()            # column-major front, lower triangle
    Lref = np.linalg.cholesky(A[:nc, :nc])
    # reference: L of the first nc columns (rows to N) and the Schur complement untouched here
    Lfull = np.linalg.cholesky(A)[:, :nc] if False else None
    # blocked reference for columns < nc
    Aw = A.copy()
    Lr = np.zeros((N, nc))
    for k in range(nc):
        Lr[k, k] = np.sqrt(Aw[k, k]); Lr[k+1:, k] = Aw[k+1:, k] / Lr[k, k]
        Aw[k+1:, k+1:] -= np.outer(Lr[k+1:, k], Lr[k+1:, k])
    dinv = {}
    X = np.zeros((nc, nc))
    npairs = (nc + 63) // 64
    for J in range(-1, npairs):
        kb = 64 * J
        w = min(64, nc - kb) if J >= 0 else 0
        kb1 = kb + w if J >= 0 else 0
        wq = min(64, nc - kb1) if kb1 < nc else 0
        w1 = min(32, wq); w2 = wq - w1
        Fin = F.copy()   # what the launch reads (everything written by earlier launches)
        # ---- role B' (one workgroup view; rows below handled the same way)
        if wq > 0:
            P = Fin[:, kb:kb + w] if w else np.zeros((N, 0))
            Lp = P[kb1:kb1 + wq, :]                       # rows of Q's pivot block in pair J's columns
            Aqq = np.tril(Fin[kb1:kb1 + wq, kb1:kb1 + wq]); Aqq = Aqq + np.tril(Aqq, -1).T
            Aqq = Aqq - Lp @ Lp.T                          # S1/S2: update of the 64x64 pivot block
            A11 = Aqq[:w1, :w1]
            L11 = np.linalg.cholesky(A11); X1 = np.linalg.inv(L11)
            dinv[kb1 // 32] = X1
            rows = np.arange(kb1 + wq, N)
            raw = Fin[rows][:, kb1:kb1 + wq]
            pv = P[rows, :]
            D = raw - pv @ Lp.T                            # D1 | D2 (lookahead with pair J)
            LQ1 = D[:, :w1] @ X1.T
            F[rows, kb1:kb1 + w1] = LQ1
            if w2 > 0:
                L21 = Aqq[w1:, :w1] @ X1.T                 # S3a
                F[kb1 + w1:kb1 + wq, kb1:kb1 + w1] = L21
                A22 = Aqq[w1:, w1:] - L21 @ L21.T          # S3b
                L22 = np.linalg.cholesky(A22); X2 = np.linalg.inv(L22)
                dinv[kb1 // 32 + 1] = X2
                D2 = D[:, w1:] - LQ1 @ L21.T               # S4
                F[rows, kb1 + w1:kb1 + wq] = D2 @ X2.T     # S5
        # ---- role A': pair J applied behind Q, own columns only
        if J >= 0:
            M0 = kb1 + wq
            if M0 < nc:
                P = Fin[:, kb:kb + w]
                U = P[M0:, :] @ P[M0:nc, :].T
                blk = F[M0:, M0:nc] - U
                F[M0:, M0:nc] = np.tril(blk, 0) if False else blk
        # ---- role C': rows R = pair J of X
        if J >= 0:
            Xin = X.copy()
            R1 = slice(kb, kb + min(32, w)); R2 = slice(kb + 32, kb + w)
            X1 = dinv[kb // 32]
            Lm = Fin  # L values final for columns < kb1
            # column tiles c0 < kb
            T1 = Lm[R1, :kb] @ Xin[:kb, :kb]
            XR1 = -X1 @ T1
            X[R1, :kb] = XR1
            X[R1, R1] = X1
            if w > 32:
                X2 = dinv[kb // 32 + 1]
                T2 = Lm[R2, :kb] @ Xin[:kb, :kb]
                L21 = Lm[R2, R1]
                X[R2, :kb] = -X2 @ (T2 + L21 @ XR1)
                X[R2, R1] = -X2 @ L21 @ X1
                X[R2, R2] = X2
    # checks
    errL = 0.0
    for p in range(0, nc, 32):
        we = min(32, nc - p)
        errL = max(errL, np.abs(F[p + we:, p:p + we] - Lr[p + we:, p:p + we]).max() if p + we < N else 0.0)
        errL = max(errL, np.abs(dinv[p // 32] - np.linalg.inv(Lr[p:p + we, p:p + we])).max())
    Xref = np.linalg.inv(np.tril(Lr[:nc, :nc]))
    errX = np.abs(np.tril(X) - Xref).max()
    return errL, errX
for N, nc in [(64, 64), (100, 64), (300, 200), (257, 129), (200, 96), (180, 97), (500, 160), (90, 33), (70, 31), (900, 900)]:
    eL, eX = run(N, nc)
    print(N, nc, "L err %.1e  X err %.1e" % (eL, eX))
