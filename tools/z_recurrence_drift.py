"""CPU-only numpy study (no GPU, no product code): how the way Z = P'K is carried through the LV loop of the kernel algorithm
(reference: src/plskern.jl:149-174) decides the orthogonality of the scores when K shrinks by orders of magnitude.

  ref     r = w - sum_i (w.P_i) R_i                      the reference's recursion (plskern.jl:156-161)
  incr    r = (K v - R (Z v)) / |K v|,  Z_i -= (P_i.zp) c'   DESIGN.md §5: what the small-state kernels do (a dots per LV)
  direct  r = (K v - R (P'K v)) / |K v|                  Z recomputed from K_new every LV (a q dots per LV)

Shape: the PLS1 case of tests/test_gpu_parity.py::test_split_small_state_matches_one_kernel_path (n = 900, p = 37, q = 1, nlv = 12).
Prints max |T'DT - diag(TT)| / max TT per form, for two row orders (= two summation orders), and |K_a|."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO


def fit(X, Y, nlv, mode, perm=None):
    n, p = X.shape; q = Y.shape[1]
    d = np.full(n, 1.0 / n)
    X = X - (d @ X); Y = Y - (d @ Y)
    if perm is not None:
        X = X[perm]; Y = Y[perm]
    K = X.T @ (d[:, None] * Y)
    P = np.zeros((p, nlv)); R = np.zeros((p, nlv)); T = np.zeros((n, nlv)); TT = np.zeros(nlv); Z = np.zeros((nlv, q)); nk_all = []
    for a in range(nlv):
        nk_all.append(np.linalg.norm(K))
        v = np.ones(1) if q == 1 else np.linalg.svd(K)[2][0]
        Kv = K @ v; nk = np.linalg.norm(Kv); w = Kv / nk
        if mode == "ref":
            r = w - R[:, :a] @ (P[:, :a].T @ w)
        elif mode == "incr":
            r = (Kv - R[:, :a] @ (Z[:a] @ v)) / nk
        else:
            r = (Kv - R[:, :a] @ ((P[:, :a].T @ K) @ v)) / nk
        t = X @ r; dt = d * t; tt = t @ dt; c = K.T @ r / tt; zp = X.T @ dt
        K = K - np.outer(zp, c)
        P[:, a] = zp / tt; R[:, a] = r; T[:, a] = t; TT[a] = tt
        Z[:a] -= np.outer(P[:, :a].T @ zp, c); Z[a] = P[:, a] @ K
    G = (T * d[:, None]).T @ T
    return np.abs(G - np.diag(TT)).max() / TT.max(), nk_all


if __name__ == "__main__":
    n, p, q, nlv = 900, 37, 1, 12
    X = CO.fill_uniform(401, n, p) + 3.0; B0 = CO.fill_uniform(402, p, q) - 0.5; Y = X @ B0 + 0.1 * CO.fill_uniform(403, n, q)
    for mode in ("ref", "incr", "direct"):
        e, nk = fit(X, Y, nlv, mode)
        e2, _ = fit(X, Y, nlv, mode, perm=np.random.default_rng(1).permutation(n))
        print(f"{mode:7s} drift {e:.2e}   rows permuted {e2:.2e}")
    print("|K_a|:", " ".join(f"{x:.1e}" for x in nk))
