"""Batched (torch, fp64, on-device) restatement of the reference's random-QP generator
generate_test_QP (interfaces/daqp-julia/test/utils.jl:3-53): same construction as
oracle.oracle.generate_qp -- H = T'T with cond(H) = kappa, random constraint rows, a chosen active
set with random multipliers so that the optimum x* is known analytically -- but N problems at a
time and already resident in HBM.  Used by bench.py and the full-size property tests; the RNG
stream differs from the numpy generator (only the distribution is the same).
"""
import torch


def _right_solve_upper(Rc, Q):
    """Q Rc^-1 for upper-triangular Rc, batched.  hipBLAS' batched trsm / getrf fail to allocate their workspace beyond n = 64
    (ROCm 7.2: HIPBLAS_STATUS_ALLOC_FAILED at n = 80 as at n = 200), so those take a column-by-column forward substitution made
    of batched mat-vecs."""
    n = Q.shape[-1]
    if n <= 64:
        return torch.linalg.solve_triangular(Rc, Q, upper=True, left=False)
    X = torch.empty_like(Q)
    for j in range(n):
        col = Q[:, :, j]
        if j:
            col = col - (X[:, :, :j] @ Rc[:, :j, j:j + 1])[:, :, 0]
        X[:, :, j] = col / Rc[:, j, j:j + 1]
    return X


def generate_batch_torch(N, n, m, ms, n_active, seed, kappa=100.0, device="cuda", chunk=None):
    """Returns dict of device tensors H (N,n,n), f (N,n), A (N,m-ms,n), bupper/blower (N,m), xref (N,n)."""
    if chunk is None:   # bounded scratch per chunk (the batched factorisations need O(chunk n^2) workspace)
        chunk = max(128, min(8192, (1 << 24) // (n * n)))
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    dd = dict(dtype=torch.float64, device=device)
    out = dict(H=torch.empty((N, n, n), **dd), f=torch.empty((N, n), **dd), A=torch.empty((N, m - ms, n), **dd),
               bupper=torch.empty((N, m), **dd), blower=torch.empty((N, m), **dd), xref=torch.empty((N, n), **dd))
    for s in range(0, N, chunk):
        B = min(chunk, N - s)
        rnd = lambda *sh: torch.rand(*sh, generator=g, **dd)
        rndn = lambda *sh: torch.randn(*sh, generator=g, **dd)
        eig = torch.empty((B, n), **dd)
        eig[:, 0] = 1.0
        eig[:, 1] = kappa
        eig[:, 2:] = 1.0 + (kappa - 1.0) * rnd(B, n - 2)
        # Haar-distributed Q by Cholesky-QR (two passes): a handful of batched kernels instead of
        # rocsolver's per-matrix Householder loop
        Q = rndn(B, n, n)
        for _ in range(2):
            Rc = torch.linalg.cholesky(Q.transpose(1, 2) @ Q, upper=True)
            Q = _right_solve_upper(Rc, Q)
        sq = eig.sqrt()
        T = sq[:, :, None] * Q.transpose(1, 2)
        Tinv = Q / sq[:, None, :]
        H = T.transpose(1, 2) @ T
        M = torch.cat([Tinv[:, :ms, :], rndn(B, m - ms, n)], dim=1)
        perm = torch.argsort(rnd(B, m), dim=1)
        n_up = torch.randint(0, n_active + 1, (B,), generator=g, device=device)
        ids_act = perm[:, :n_active]
        is_up = torch.arange(n_active, device=device)[None, :] < n_up[:, None]
        sgn = torch.where(is_up, 1.0, -1.0).to(torch.float64)
        lam = rnd(B, n_active)
        Ma = sgn[:, :, None] * torch.gather(M, 1, ids_act[:, :, None].expand(B, n_active, n))
        u = -(Ma.transpose(1, 2) @ lam[:, :, None])[:, :, 0]
        da = (Ma @ u[:, :, None])[:, :, 0]
        Mu = (M @ u[:, :, None])[:, :, 0]
        dupper = Mu + 0.01 + rnd(B, m)
        dlower = Mu - (0.01 + rnd(B, m))
        gap = 0.01 + rnd(B, n_active)
        du_act = torch.where(is_up, da, -da + gap)
        dl_act = torch.where(is_up, da - gap, -da)
        dupper.scatter_(1, ids_act, du_act)
        dlower.scatter_(1, ids_act, dl_act)
        v = rndn(B, n)
        f = (T.transpose(1, 2) @ v[:, :, None])[:, :, 0]
        x = (Tinv @ (u - v)[:, :, None])[:, :, 0]
        Mv = (M @ v[:, :, None])[:, :, 0]
        out["H"][s:s + B] = 0.5 * (H + H.transpose(1, 2))
        out["f"][s:s + B] = f
        out["A"][s:s + B] = M[:, ms:, :] @ T
        out["bupper"][s:s + B] = dupper - Mv
        out["blower"][s:s + B] = dlower - Mv
        out["xref"][s:s + B] = x
    return out
