"""Second, independent restatement of the reference hot path in pure Python / numpy-float32
scalars (small cases only).  Written from the reference source, not from the C oracle, so that a
transcription slip in oracle/vsom_oracle.c shows up as a disagreement.  Citations are into
/root/reference (src/Som.cpp, src/Transformation.cpp, src/SomIndex.cpp).
"""
import math

import numpy as np

f32 = np.float32
STANDARD, MEDIAN, CLR = 0, 1, 2
EXPONENTIAL, INVERSE_PROPORTIONAL = 0, 1
SIZE_MAX = (1 << 64) - 1


def length(tr, J):
    return J * (J - 1) if tr == CLR else J          # Transformation.cpp:162-165


def _pairs(J):
    return [(i, j) for i in range(J) for j in range(i + 1, J)]   # Transformation.cpp:95-101


def comparer(tr, v, m):
    v = np.asarray(v, f32)
    m = np.asarray(m, f32)
    if tr == CLR:
        P = m.size // 2
        out = np.empty(P, f32)
        for p, (i, j) in enumerate(_pairs(v.size)):
            t = f32(m[p] * v[i])
            t = f32(t + m[P + p])
            out[p] = f32(t - v[j])                  # Transformation.cpp:104
        return out
    return (m - v).astype(f32)                       # Transformation.cpp:8,46


def _sign(a):
    if np.isnan(a):
        return f32(a)
    return f32(int(a > 0) - int(a < 0))


def stepper(tr, v, m):
    v = np.asarray(v, f32)
    m = np.asarray(m, f32)
    if tr == CLR:
        P = m.size // 2
        out = np.empty(2 * P, f32)
        for p, (i, j) in enumerate(_pairs(v.size)):
            inner = f32(f32(f32(m[p] * v[i]) + m[P + p]) - v[j])   # :129
            m2 = f32(f32(-2.0) * inner)                            # :135-136
            out[p] = f32(m2 * v[i])
            out[P + p] = m2
        return out
    d = (v - m).astype(f32)                          # :12
    if tr == MEDIAN:
        return np.array([_sign(a) for a in d], f32)  # :50
    return d


def dot_self(r):
    """Eigen 3.4 redux, LinearVectorizedTraversal, Packet4f."""
    r = np.asarray(r, f32)
    n = r.size
    if n == 0:
        return f32(0)
    p = (r * r).astype(f32)
    a2, a1 = (n // 8) * 8, (n // 4) * 4
    if a1:
        p0 = p[0:4].copy()
        if a1 > 4:
            p1 = p[4:8].copy()
            for idx in range(8, a2, 8):
                p0 = (p0 + p[idx:idx + 4]).astype(f32)
                p1 = (p1 + p[idx + 4:idx + 8]).astype(f32)
            p0 = (p0 + p1).astype(f32)
            if a1 > a2:
                p0 = (p0 + p[a2:a2 + 4]).astype(f32)
        res = f32(f32(p0[0] + p0[2]) + f32(p0[1] + p0[3]))
        for idx in range(a1, n):
            res = f32(res + p[idx])
        return res
    res = p[0]
    for idx in range(1, n):
        res = f32(res + p[idx])
    return f32(res)


def nbh(cx, cy, bx, by, sigma):
    if sigma > 1.0:                                   # Som.cpp:955-963
        dx, dy = float(cx) - float(bx), float(cy) - float(by)
        return math.exp(-(dx * dx / 2.0 / sigma / sigma + dy * dy / 2.0 / sigma / sigma))
    return 1.0 if (cx == bx and cy == by) else 0.0    # :967-974


class Som:
    def __init__(self, W, H, J, tr=STANDARD):
        self.W, self.H, self.J, self.tr = W, H, J, tr
        self.D = length(tr, J)
        N = W * H
        self.map = np.zeros((N, self.D), f32)
        self.sigma = np.zeros((N, self.D), f32)
        self.S = np.zeros((N, self.D), f32)
        self.weight = np.zeros(N, f32)
        self.hits = np.zeros(N, np.uint64)

    def somindex(self, idx):                          # SomIndex.cpp:13-18
        x = idx % self.W
        return x, (idx - x) // self.H

    def dist(self, node, v):                          # Som.cpp:124-141
        return dot_self(comparer(self.tr, v, self.map[node]))

    def find_bmu(self, v):                            # Som.cpp:291-309
        best, bi = self.dist(0, v), 0
        for i in range(self.W * self.H):
            d = self.dist(i, v)
            if d < best:
                best, bi = d, i
        return bi

    def find_local_bmu(self, v, last):                # Som.cpp:335-454
        W, H = self.W, self.H
        M = 1 << 64
        lastBMU = last
        minDist, minIndex = self.dist(lastBMU, v), lastBMU
        fx = [SIZE_MAX, 0, 1, 1, 1, 0, SIZE_MAX, SIZE_MAX]
        fy = [1, 1, 1, 0, SIZE_MAX, SIZE_MAX, SIZE_MAX, 0]
        lastMeasured = lastBMU
        while True:
            lmX, lmY = lastMeasured % W, lastMeasured // W
            lbX, lbY = lastBMU % W, lastBMU // W
            if lastMeasured == lastBMU:
                for i in range(8):
                    cx = min((lmX + fx[i]) % M, W - 1)
                    cy = min((lmY + fy[i]) % M, H - 1)
                    d = self.dist(cy * W + cx, v)
                    if d < minDist:
                        minDist, minIndex = d, cy * W + cx
                if minIndex == lastBMU:
                    return minIndex
                lastMeasured = minIndex
            else:
                if (lmX - lbX) % M:
                    for i in (-1, 0, 1):
                        cx = min((lmX + lmX - lbX) % M, W - 1)
                        cy = min((lmY + i) % M, H - 1)
                        d = self.dist(cy * W + cx, v)
                        if d < minDist:
                            minDist, minIndex = d, cy * W + cx
                if (lmY - lbY) % M:
                    if (lmX - lbX) % M > 0:
                        startX, endX = SIZE_MAX, 0
                    else:
                        startX, endX = SIZE_MAX, 1
                    i = startX
                    while i < endX + 1:               # SIZE_MAX < small: never (Som.cpp:426)
                        raise AssertionError("unreachable in the reference")
                if minIndex == lastMeasured:
                    return minIndex
                lastBMU, lastMeasured = lastMeasured, minIndex

    def batch_epoch(self, X, lastbmu, sigma, is_first):   # Som.cpp:756-879
        X = np.asarray(X, f32)
        B = X.shape[0]
        mse = f32(0)
        for s in range(B):
            idx = self.find_bmu(X[s]) if is_first else self.find_local_bmu(X[s], int(lastbmu[s]))
            lastbmu[s] = idx
            self.hits[idx] += 1
            res = comparer(self.tr, X[s], self.map[idx])
            mse = f32(mse + f32(dot_self(res) / f32(B)))
        newmap = np.zeros_like(self.map)
        for node in range(self.W * self.H):
            cx, cy = self.somindex(node)
            Wsum = f32(0)
            M = np.zeros(self.D, f32)
            S = np.zeros(self.D, f32)
            for j in range(B):
                bx, by = self.somindex(int(lastbmu[j]))
                w = f32(nbh(cx, cy, bx, by, sigma))
                Wsum = f32(Wsum + w)
                with np.errstate(all="ignore"):
                    last = M.copy()
                    delta = stepper(self.tr, X[j], M)
                    c = f32(w / Wsum)
                    M = (M + (c * delta).astype(f32)).astype(f32)
                    d2 = stepper(self.tr, X[j], last)
                    S = (S + ((w * d2).astype(f32) * delta).astype(f32)).astype(f32)
            newmap[node] = M
            with np.errstate(all="ignore"):
                self.sigma[node] = np.sqrt((S / Wsum).astype(f32)).astype(f32)
            self.weight[node] = Wsum
        self.map[...] = newmap    # phase 2 never reads the map, so in-place == the reference
        return mse

    def train_single(self, v, eta, sigma, last, fn):      # Som.cpp:885-947
        v = np.asarray(v, f32)
        W, H = self.W, self.H
        bmu = self.find_bmu(v) if sigma > 1 else self.find_local_bmu(v, last)
        bx, by = bmu % W, bmu // W
        sx = int(max(float(bx) - 2.5 * sigma, 0.0))
        sy = int(max(float(by) - 2.5 * sigma, 0.0))
        ex = int(min(float(bx) + 2.5 * sigma, float(W)))
        ey = int(min(float(by) + 2.5 * sigma, float(H)))
        with np.errstate(all="ignore"):
            for j in range(sy, ey):
                for i in range(sx, ex):
                    n = j * W + i
                    delta = stepper(self.tr, v, self.map[n])
                    h = nbh(i, j, bx, by, sigma)
                    if fn == EXPONENTIAL:
                        self.weight[n] = f32(self.weight[n] + f32(h * eta))
                        self.map[n] = (self.map[n] + (f32(h * eta) * delta).astype(f32)).astype(f32)
                    else:
                        self.weight[n] = f32(self.weight[n] + f32(h))
                        tw = 1.0 if self.weight[n] == 0 else h / float(self.weight[n])
                        self.map[n] = (self.map[n] + (f32(tw) * stepper(self.tr, v, self.map[n])).astype(f32)).astype(f32)
                    tw2 = 0.000001 if self.weight[n] == 0 else float(self.weight[n])
                    d2 = stepper(self.tr, v, self.map[n])
                    self.S[n] = (self.S[n] + (f32(h) * (delta * d2).astype(f32)).astype(f32)).astype(f32)
                    self.sigma[n] = np.sqrt(np.abs((self.S[n] / f32(tw2)).astype(f32))).astype(f32)
        res = comparer(self.tr, v, self.map[bmu])
        return bmu, res, dot_self(res), by * W + bx
