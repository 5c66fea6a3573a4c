"""numpy prototype of the proximal-ALM + semismooth-Newton finish for general convex QPs (OSQP form)."""
import numpy as np

def ruiz(P, q, A, l, u, iters=4):
    n, m = len(q), len(l)
    D = np.ones(n); E = np.ones(m); c = 1.0
    P = P.copy(); q = q.copy(); A = A.copy()
    for _ in range(iters):
        cn = np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0) if m else 0)
        cn = np.where(cn < 1e-4, 1.0, np.minimum(cn, 1e4))
        Dt = 2.0 ** -(np.frexp(cn)[1] >> 1)
        rn = np.abs(A).max(axis=1)
        rn = np.where(rn < 1e-4, 1.0, np.minimum(rn, 1e4))
        Et = 2.0 ** -(np.frexp(rn)[1] >> 1)
        P = Dt[:, None] * P * Dt[None, :]; q = Dt * q; A = Et[:, None] * A * Dt[None, :]
        D *= Dt; E *= Et
        cm = np.abs(P).sum(axis=0).mean(); qn = np.abs(q).max(); qn = 1.0 if qn < 1e-4 else min(qn, 1e4)
        ct = max(cm, qn); ct = 1.0 if ct < 1e-4 else min(ct, 1e4)
        ct = 2.0 ** (np.frexp(1.0 / ct)[1] - 1)
        P *= ct; q *= ct; c *= ct
    return P, q, A, l * E, u * E, D, E, c

def nalm(P, q, A, l, u, x0=None, y0=None, gamma=1e4, mu0=1e1, mu_max=1e4, tol=1e-10, max_newton=200, verbose=False):
    n, m = len(q), len(l)
    x = np.zeros(n) if x0 is None else x0.copy()
    y = np.zeros(m) if y0 is None else y0.copy()
    mu = np.full(m, mu0)
    eq = (u - l) < 1e-4
    mu[eq] *= 1e2
    xh = x.copy()
    newton = 0; factors = 0
    Jprev = None; K = None
    pri_prev = None
    for outer in range(60):
        # inner: semismooth Newton on phi
        for inner in range(50):
            s = A @ x + y / mu
            proj = np.clip(s, l, u)
            r = mu * (s - proj)
            g = P @ x + q + (x - xh) / gamma + A.T @ r
            gn = np.abs(g).max()
            gscale = 1 + max(np.abs(P @ x).max(), np.abs(q).max(), np.abs(A.T @ r).max())
            if gn <= 0.1 * tol * gscale:
                break
            J = (s < l) | (s > u)
            if Jprev is None or not np.array_equal(J, Jprev):
                K = P + np.eye(n) / gamma + (A[J].T * mu[J]) @ A[J]
                L = np.linalg.cholesky(K); factors += 1; Jprev = J
            d = -np.linalg.solve(L.T, np.linalg.solve(L, g))
            newton += 1
            # exact line search on the piecewise quadratic: phi'(t) = g.d + t d'(P+I/gamma)d + sum_i mu_i [ (s_i+t dl_i - proj(..)) - (s_i-proj_i) ] dl_i
            dl = A @ d
            a0 = g @ d; a1 = d @ (P @ d) + (d @ d) / gamma
            def dphi(t):
                st = s + t * dl
                return a0 + t * a1 + np.sum(mu * ((st - np.clip(st, l, u)) - (s - proj)) * dl)
            # breakpoints
            with np.errstate(divide='ignore', invalid='ignore'):
                tl = (l - s) / dl; tu = (u - s) / dl
            bps = np.concatenate([tl, tu]); bps = bps[np.isfinite(bps) & (bps > 0)]
            bps = np.sort(bps)
            tlo, thi = 0.0, None
            flo = dphi(0.0)
            for tb in bps:
                fb = dphi(tb)
                if fb >= 0: thi = tb; fhi = fb; break
                tlo, flo = tb, fb
            if thi is None:
                # beyond the last breakpoint phi' is linear with slope from current active set
                st = s + (tlo + 1.0) * dl
                f1 = dphi(tlo + 1.0)
                slope = f1 - flo
                t = tlo - flo / slope if slope > 0 else 1.0
            else:
                t = tlo - flo * (thi - tlo) / (fhi - flo) if fhi > flo else tlo
            x = x + t * d
        s = A @ x + y / mu
        ynew = mu * (s - np.clip(s, l, u))
        Ax = A @ x
        pri = np.abs(Ax - np.clip(Ax, l, u)).max()
        dua = np.abs(P @ x + q + A.T @ ynew).max()
        dy = ynew - y
        y = ynew; xh = x.copy()
        ndy = np.abs(dy).max()
        if ndy > 1e-4:
            big = 1e20
            v = np.where(u > big, np.where(l < -big, 0.0, np.minimum(dy, 0)), np.where(l < -big, np.maximum(dy, 0), dy))
            lhs = np.sum(np.where(v > 0, u * v, 0) + np.where(v < 0, l * v, 0))
            if np.abs(v).max() > 1e-4 * 0 and lhs < -1e-6 * np.abs(v).max() and np.abs(A.T @ v).max() < 1e-6 * np.abs(v).max():
                return x, y, -3, newton, factors, outer + 1
        if verbose: print(outer, inner, pri, dua, newton)
        if pri <= tol * (1 + np.abs(Ax).max()) and dua <= tol * (1 + max(np.abs(P @ x).max(), np.abs(q).max(), np.abs(A.T @ y).max())):
            xp, yp, ok = polish(P, q, A, l, u, x, y)
            if ok: return xp, yp, 1, newton, factors, outer + 1
            return x, y, 2, newton, factors, outer + 1
        if pri_prev is not None and pri > 0.1 * pri_prev:
            mu = np.minimum(mu * 10, mu_max); Jprev = None
        pri_prev = pri
        if newton > max_newton: break
    return x, y, 0, newton, factors, outer + 1

def polish(P, q, A, l, u, x, y, idl=1e10, steps=4):
    """active set frozen from the ALM duals (exactly zero on inactive rows); equality-constrained QP by the
    method of multipliers with penalty idl, `steps` refinement steps; returns x, nu, ok (full KKT valid)."""
    n = len(q)
    J = y != 0
    rhsJ = np.where(y < 0, l, u)[J]
    AJ = A[J]
    reg = np.where(np.diag(P) > 0, 0.0, 1e-7)
    K = P + np.diag(reg) + idl * AJ.T @ AJ
    try:
        L = np.linalg.cholesky(K)
    except np.linalg.LinAlgError:
        return x, y, False
    xp = x.copy(); nu = y[J].copy()
    for _ in range(steps):
        e = rhsJ - AJ @ xp
        rhs = -q - P @ xp + AJ.T @ (e * idl - nu)
        dx = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        nu = nu + (AJ @ dx - e) * idl
        xp = xp + dx
    Ax = A @ xp
    tol = 1e-9 * (1 + np.abs(Ax))
    ok = np.all(Ax >= l - tol) and np.all(Ax <= u + tol)
    yy = np.zeros_like(y); yy[J] = nu
    # sign: y<0 at lower bound, y>0 at upper (osqp convention: Px+q+A'y=0)
    eq = (u - l) < 1e-9
    bad = (~eq) & (((np.where(y < 0, 1, 0) == 1) & (yy > 1e-9 * (1 + np.abs(yy)))) | ((y > 0) & (yy < -1e-9 * (1 + np.abs(yy)))))
    ok = ok and not bad.any()
    g = P @ xp + q + A.T @ yy
    mag = np.abs(P @ xp) + np.abs(q) + np.abs(A.T) @ np.abs(yy)
    ok = ok and np.all(np.abs(g) <= 1e-9 * mag + 1e-300)
    return xp, yy, ok

def to_osqp(nv, nc, Hd, c, Acm, b, lb, ub, be):
    P = np.diag(2.0 * Hd); q = c.copy()
    A = np.vstack([Acm.reshape(nv, nc).T, np.eye(nv)])
    l = np.concatenate([b, lb]); u = np.concatenate([np.where(be.astype(bool), b, 1e30), ub])
    return P, q, A, l, u

if __name__ == "__main__":
    import sys, time
    sys.path.insert(0, '/root/repo/tests')
    import oracle_lib as O
    model, variant = O.CONFIGS[5]
    o = O.default_options(model, variant)
    B = 400
    xs, us = O.make_batch(5, B)
    ua, rl, rc = O.filter_batch(model, variant, o, xs, us, O.SOLVER_EXACT)
    A_, b_, code, _ = O.assemble_batch(model, variant, o, xs)
    d = O.dims(model, variant, o)
    errs = []; nts = []; fcs = []; fails = 0
    for i in range(B):
        Hd, c, lb, ub, be = O.qp_static(model, variant, o, us[i])
        P, q, A, l, u = to_osqp(d.nv, d.nc, Hd, c, A_[i], b_[i], lb, ub, be)
        Ps, qs, As, ls, us_, D, E, cc = ruiz(P, q, A, l, u, 4)
        x, y, st, nt, fc, outer = nalm(Ps, qs, As, ls, us_)
        xo = D * x
        if st != 1: fails += 1
        errs.append(abs(min(max(xo[0], o.lb[0]), o.ub[0]) - ua[i, 0])); nts.append(nt); fcs.append(fc)
    print("C5 full 18x12: fails", fails, "max err", max(errs), "newton mean/max", np.mean(nts), max(nts), "factor mean/max", np.mean(fcs), max(fcs))
