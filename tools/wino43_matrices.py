"""Exact F(4x4,3x3) Winograd matrices for the symmetric point set {0, +-p, +-q, inf} (Toom-Cook; rationals).  A^T is the
Vandermonde matrix, G = Vandermonde / N_j with N_j chosen so that B^T is what the identity
    sum_j A^T[i][j] G[j][k] B^T[j][l] = [l == i + k]
forces for that G.  Prints the three matrices and the butterfly coefficients the kernels use.
    python tools/wino43_matrices.py [p q]          (default 3/4 3/2: the lowest fp32 error of the sets tried, tools/wino43_error.py)"""
import sys
from fractions import Fraction as Fr

def solve(A, b):                     # least-squares-free: A is (rows x n) with full column rank and consistent rhs
    n = len(A[0])
    M = [row[:] + [bb] for row, bb in zip(A, b)]
    piv = []
    rix = 0
    for c in range(n):
        pr = next((i for i in range(rix, len(M)) if M[i][c] != 0), None)
        if pr is None:
            continue
        M[rix], M[pr] = M[pr], M[rix]
        M[rix] = [v / M[rix][c] for v in M[rix]]
        for i in range(len(M)):
            if i != rix and M[i][c] != 0:
                M[i] = [vi - M[i][c] * vr for vi, vr in zip(M[i], M[rix])]
        piv.append(c); rix += 1
    assert all(all(v == 0 for v in row[:-1]) and row[-1] == 0 for row in M[rix:]), "inconsistent"
    x = [Fr(0)] * n
    for i, c in enumerate(piv):
        x[c] = M[i][-1]
    return x



def matrices(p=Fr(3, 4), q=Fr(3, 2)):
    """(A^T 4x6, G 6x3, B^T 6x6) as Fractions for the points 0, +-p, +-q, inf."""
    pts = [Fr(0), p, -p, q, -q]
    m, r, a = 4, 3, 6
    AT = [[(pt ** i if i else Fr(1)) for pt in pts] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    # G scaled by the Lagrange denominators N_j = prod_{k != j} (a_j - a_k) (finite points), which makes B^T polynomial coefficients
    N = []
    for j, aj in enumerate(pts):
        d = Fr(1)
        for k, ak in enumerate(pts):
            if k != j:
                d *= (aj - ak)
        N.append(d)
    G = [[(pt ** k if k else Fr(1)) / N[j] for k in range(r)] for j, pt in enumerate(pts)] + [[Fr(0), Fr(0), Fr(1)]]
    BT = [[Fr(0)] * a for _ in range(a)]
    for l in range(a):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append([AT[i][j] * G[j][k] for j in range(a)]); rhs.append(Fr(1 if l == i + k else 0))
        col = solve(rows, rhs)
        for j in range(a):
            BT[j][l] = col[j]
    return AT, G, BT


if __name__ == "__main__":
    p, q = (Fr(sys.argv[1]), Fr(sys.argv[2])) if len(sys.argv) > 2 else (Fr(3, 4), Fr(3, 2))
    AT, G, BT = matrices(p, q)
    fmt = lambda M: "\n".join("  [" + ", ".join(f"{str(v):>9s}" for v in row) + "]" for row in M)
    print("points 0, +-%s, +-%s, inf" % (p, q))
    print("A^T =\n" + fmt(AT)); print("G =\n" + fmt(G)); print("B^T =\n" + fmt(BT))
    print("float B^T:", [[float(v) for v in row] for row in BT])
    print("float G:", [[float(v) for v in row] for row in G])
