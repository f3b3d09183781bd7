"""CPU experiment (numpy, no GPU): error of Winograd F(4x4,3x3) against F(2x2,3x3) for one 3x3 layer when the transform-domain operands carry
22 significant bits (hi + lo IEEE halves) and everything else is fp32 -- the arithmetic of the split-half Winograd kernels -- against an fp64
direct convolution.  F(4,3) does 2.25 transform-domain multiplies per output instead of 4 (and streams 2.25 instead of 4 weight positions per
output pixel), but its transforms carry constants up to 8, which amplifies rounding.  Usage: python tools/winograd_f43_numerics.py [C] [H]"""
import sys
import numpy as np

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 48          # multiple of 4
rng = np.random.default_rng(0)

# F(2,3) (Lavin & Gray) and F(4,3) transform matrices
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def split22(v):
    """fp32 -> hi + lo IEEE halves (22 significant bits), as fp32; values assumed inside the half range after scaling."""
    v = v.astype(np.float32)
    hi = v.astype(np.float16).astype(np.float32)
    lo = (v - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def winograd(x, w, BT, G, AT, m):
    """x: (C, H, H) fp32 zero-padded by the caller, w: (K, C, 3, 3).  Tiles of m x m outputs; transforms in fp32, products as
    hi*hi + hi*lo + lo*hi accumulated in fp32 (the order of the kernels up to associativity), output transform in fp32."""
    a = m + 2
    K = w.shape[0]
    U = np.einsum('ij,kcjl,ml->imkc', G, w.astype(np.float64), G)                     # pack time: fp64, then scaled + split
    s = 2.0 ** (14 - np.floor(np.log2(np.abs(U).max())) - 1)
    Uh, Ul = split22(U * s)
    n = (x.shape[1] - 2) // m
    out = np.zeros((K, n * m, n * m), np.float32)
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    for ty in range(n):
        for tx in range(n):
            d = x[:, ty * m:ty * m + a, tx * m:tx * m + a].astype(np.float32)
            V = np.einsum('ij,cjl,ml->imc', BT32, d, BT32).astype(np.float32)           # fp32 input transform
            Vh, Vl = split22(V)
            M = (np.einsum('imkc,imc->imk', Uh, Vh) + np.einsum('imkc,imc->imk', Ul, Vh) + np.einsum('imkc,imc->imk', Uh, Vl)).astype(np.float32)
            Y = np.einsum('ij,jlk,ml->imk', AT32, M, AT32).astype(np.float32) / np.float32(s)
            out[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = Y.transpose(2, 0, 1)
    return out


def toom(points, m=4, r=3):
    """Transforms of F(m, r) for the finite interpolation points `points` plus infinity (Toom-Cook).  AT and G follow from the points;
    BT is solved from the bilinear identity AT [(G g) * (BT d)] = valid correlation of d with g (reproduces Lavin & Gray's matrices for
    (0, 1, -1, 2, -2))."""
    n = m + r - 1
    p = np.array(points, np.float64)
    AT, G = np.zeros((m, n)), np.zeros((n, r))
    for j in range(n - 1):
        AT[:, j] = p[j] ** np.arange(m)
        G[j] = p[j] ** np.arange(r) / np.prod([p[j] - p[l] for l in range(n - 1) if l != j])
    AT[m - 1, n - 1] = G[n - 1, r - 1] = 1.0
    g0 = np.random.default_rng(1)
    rows, rhs = [], []
    for _ in range(80):
        d, g = g0.standard_normal(n), g0.standard_normal(r)
        for i in range(m):
            rows.append(np.outer(AT[i] * (G @ g), d).ravel())
            rhs.append(np.dot(d[i:i + r], g))
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
    BT[np.abs(BT) < 1e-12] = 0.0
    return BT, G, AT


x = rng.standard_normal((C, H, H))
x = x / (1 + np.exp(-x))                                   # swish of N(0,1): what a GN-swish prologue hands the conv
w = rng.standard_normal((C, C, 3, 3)) * np.sqrt(2.0 / (9 * C))
xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
ref = np.zeros((C, H, H))
for ky in range(3):
    for kx in range(3):
        ref += np.einsum('kc,chw->khw', w[:, :, ky, kx], xp[:, ky:ky + H, kx:kx + H])
cases = [('F(2x2,3x3)', (BT2, G2, AT2, 2)), ('F(4x4,3x3) points 0 +-1 +-2 inf (Lavin & Gray)', (BT4, G4, AT4, 4))]
for pts in ((0, 1, -1, 0.5, -0.5), (0, 1, -1, 0.5, -2), (0, 1, -1, 2, -0.5), (0, 0.5, -0.5, 2, -2), (0, 1, -1, 1.5, -1.5)):
    cases.append((f'F(4x4,3x3) points {pts} inf', toom(pts) + (4,)))
for name, (BT, G, AT, m) in cases:
    y = winograd(xp, w, BT, G, AT, m)
    e = np.abs(y - ref)
    print(f'{name}: max err {e.max():.2e}  mean err {e.mean():.2e}  (output max {np.abs(ref).max():.2f}, C = {C}, {H}x{H})')
