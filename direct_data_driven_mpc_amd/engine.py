"""Batched Data-Driven MPC QP engine: Python host side over the C ABI.

`BatchedDDMPC` owns one `ddmpc_handle` = one batch of independent controller
instances that share the controller parameters and differ in their data
trajectories (u_d, y_d) and past windows.  This is the data-parallel axis the
reference does not have (one `DirectDataDrivenMPCController` = one instance,
direct_data_driven_mpc_controller.py:22).

Buffers may be numpy arrays (host; the library stages them, calls are
synchronous) or torch CUDA tensors (device resident; calls are asynchronous on
the current torch stream).  torch is plumbing only and is imported lazily.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib as L


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _weights(W, size: int, name: str) -> Tuple[int, np.ndarray]:
    """Accept scalar / 1-D diagonal / 2-D matrix; return (kind, values).  A 2-D matrix that is diagonal is
    passed as its diagonal, anything else as the dense matrix (symmetry and definiteness are checked by
    `ddmpc_create`)."""
    W = np.asarray(W, dtype=np.float64)
    if W.ndim == 0:
        return L.WEIGHT_SCALAR, W.reshape(1).copy()
    if W.ndim == 1:
        if W.shape[0] != size:
            raise ValueError("%s diagonal must have %d entries" % (name, size))
        d = W
    elif W.ndim == 2:
        if W.shape != (size, size):
            raise ValueError("%s must be %dx%d" % (name, size, size))
        d = np.diag(W)
        if np.any(W != np.diag(d)):
            return L.WEIGHT_DENSE, np.ascontiguousarray(W, dtype=np.float64)
    else:
        raise ValueError("%s has too many dimensions" % name)
    if np.all(d == d[0]):
        return L.WEIGHT_SCALAR, np.array([d[0]], dtype=np.float64)
    return L.WEIGHT_DIAG, np.ascontiguousarray(d, dtype=np.float64)


def _expand_weight(kind: int, v: np.ndarray, size: int, target: int) -> np.ndarray:
    """Bring a weight to the representation `target` (both weights travel in the same kind)."""
    if kind == target:
        return v
    d = np.full(size, v[0]) if kind == L.WEIGHT_SCALAR else v
    return d if target == L.WEIGHT_DIAG else np.ascontiguousarray(np.diag(d))


class BatchedDDMPC:
    def __init__(self, n: int, m: int, p: int, L_: int, N: int, Q, R, u_s, y_s, batch: int,
                 controller_type: int = L.ROBUST, slack_type: int = L.SLACK_NONE,
                 eps_max: Optional[float] = None, lamb_alpha: Optional[float] = None,
                 lamb_sigma: Optional[float] = None, c: Optional[float] = None,
                 use_terminal_constraint: bool = True, device: int = 0, max_iter: int = 0,
                 gram_mode: int = L.GRAM_AUTO):
        self._lib = L.load()
        self.n, self.m, self.p, self.L, self.N, self.batch = int(n), int(m), int(p), int(L_), int(N), int(batch)
        self.device = int(device)
        self.controller_type, self.slack_type = int(controller_type), int(slack_type)
        wk_q, q = _weights(Q, self.p * self.L, "Q")
        wk_r, r = _weights(R, self.m * self.L, "R")
        if wk_q != wk_r:      # mixed: bring both to the richer representation
            target = max(wk_q, wk_r)
            q = _expand_weight(wk_q, q, self.p * self.L, target)
            r = _expand_weight(wk_r, r, self.m * self.L, target)
            wk_q = wk_r = target
        self._q, self._r = q, r
        self._us = np.ascontiguousarray(np.asarray(u_s, dtype=np.float64).reshape(-1))
        self._ys = np.ascontiguousarray(np.asarray(y_s, dtype=np.float64).reshape(-1))
        if self._us.size != self.m or self._ys.size != self.p:
            raise ValueError("u_s / y_s must have m / p entries")
        prm = L.Params()
        prm.struct_size = C.sizeof(L.Params)
        prm.m, prm.p, prm.n, prm.L, prm.N = self.m, self.p, self.n, self.L, self.N
        prm.controller_type, prm.slack_type = self.controller_type, self.slack_type
        prm.use_terminal_constraint = 1 if use_terminal_constraint else 0
        prm.weight_kind = wk_q
        prm.Q = self._q.ctypes.data_as(L.c_double_p)
        prm.R = self._r.ctypes.data_as(L.c_double_p)
        prm.eps_max = float(eps_max) if eps_max is not None else 0.0
        prm.lamb_alpha = float(lamb_alpha) if lamb_alpha is not None else 0.0
        prm.lamb_sigma = float(lamb_sigma) if lamb_sigma is not None else 0.0
        prm.c = float(c) if c is not None else 0.0
        prm.u_s = self._us.ctypes.data_as(L.c_double_p)
        prm.y_s = self._ys.ctypes.data_as(L.c_double_p)
        prm.max_iter = int(max_iter)
        prm.gram_mode = int(gram_mode)
        self._h = C.c_void_p()
        L.check(self._lib.ddmpc_create(C.byref(prm), self.batch, self.device, C.byref(self._h)))
        self._keep = {}

    # ---- lifetime ------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ddmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- helpers -----------------------------------------------------------
    def _ptr(self, x, shape, name, dtype=np.float64, writable=False):
        """Return (pointer, mem, keepalive) for a numpy array or a torch CUDA tensor."""
        if _is_torch(x):
            import torch
            want = {np.float64: torch.float64, np.int32: torch.int32}[dtype]
            if not x.is_cuda or x.dtype != want or not x.is_contiguous():
                raise ValueError("%s must be a contiguous CUDA tensor of dtype %s" % (name, want))
            if tuple(x.shape) != tuple(shape):
                raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(x.shape)))
            return C.c_void_p(x.data_ptr()), L.MEM_DEVICE, x
        a = np.asarray(x)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(a.shape)))
        if writable:
            if a.dtype != dtype or not a.flags.c_contiguous or not a.flags.writeable:
                raise ValueError("%s must be a writable C-contiguous %s array" % (name, np.dtype(dtype)))
        else:
            a = np.ascontiguousarray(a, dtype=dtype)
        return C.c_void_p(a.ctypes.data), L.MEM_HOST, a

    def _use_torch_stream(self):
        import torch
        self._lib.ddmpc_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    # ---- API ---------------------------------------------------------------
    def set_data(self, u_d, y_d) -> None:
        """u_d [batch,N,m], y_d [batch,N,p] (controller.py:177-179)."""
        pu, mu, ku = self._ptr(u_d, (self.batch, self.N, self.m), "u_d")
        py, my, ky = self._ptr(y_d, (self.batch, self.N, self.p), "y_d")
        if mu != my:
            raise ValueError("u_d and y_d must live in the same memory space")
        if mu == L.MEM_DEVICE:
            self._use_torch_stream()
        L.check(self._lib.ddmpc_set_data(self._h, pu, py, mu))
        self._keep["data"] = (ku, ky)

    def solve(self, u_past, y_past, u_opt=None, cost=None, status=None, iters=None, warm: bool = False):
        """One QP solve per instance.  Returns (u_opt, cost, status, iters).

        warm=False: cold solve (Hankel -> Gram -> KKT -> Cholesky -> solve), `ddmpc_solve`.
        warm=True : `ddmpc_step`, the affine law prepared once per data set; with slack CONVEX the instances
                    whose affine iterate leaves the slack box are re-solved cold inside the same call."""
        B = self.batch
        dev = _is_torch(u_past)
        if u_opt is None:
            if dev:
                import torch
                kw = dict(device=u_past.device)
                u_opt = torch.empty((B, self.L * self.m), dtype=torch.float64, **kw)
                cost = torch.empty((B,), dtype=torch.float64, **kw)
                status = torch.empty((B,), dtype=torch.int32, **kw)
                iters = torch.empty((B,), dtype=torch.int32, **kw)
            else:
                u_opt = np.empty((B, self.L * self.m))
                cost = np.empty((B,))
                status = np.empty((B,), dtype=np.int32)
                iters = np.empty((B,), dtype=np.int32)
        p1, m1, k1 = self._ptr(u_past, (B, self.n * self.m), "u_past")
        p2, m2, k2 = self._ptr(y_past, (B, self.n * self.p), "y_past")
        p3, m3, k3 = self._ptr(u_opt, (B, self.L * self.m), "u_opt", writable=True)
        p4, m4, k4 = self._ptr(cost, (B,), "cost", writable=True)
        p5, m5, k5 = self._ptr(status, (B,), "status", dtype=np.int32, writable=True)
        if iters is not None:
            p6, m6, k6 = self._ptr(iters, (B,), "iters", dtype=np.int32, writable=True)
        else:
            p6, m6, k6 = C.c_void_p(), m1, None
        if len({m1, m2, m3, m4, m5, m6}) != 1:
            raise ValueError("all solve buffers must live in the same memory space")
        if m1 == L.MEM_DEVICE:
            self._use_torch_stream()
        fn = self._lib.ddmpc_step if warm else self._lib.ddmpc_solve
        L.check(fn(self._h, p1, p2, p3, p4, p5, p6, m1))
        self._keep["solve"] = (k1, k2, k3, k4, k5, k6)
        return u_opt, cost, status, iters

    def solve_from_host(self, u_d, y_d, u_past, y_past):
        """`set_data` + `solve` for host arrays in one pipelined call (uploads overlapped with the solves)."""
        B = self.batch
        arrs = []
        for a, shape, name in ((u_d, (B, self.N, self.m), "u_d"), (y_d, (B, self.N, self.p), "y_d"),
                               (u_past, (B, self.n * self.m), "u_past"), (y_past, (B, self.n * self.p), "y_past")):
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != shape:
                raise ValueError("%s must have shape %s" % (name, (shape,)))
            arrs.append(a)
        u_opt = np.empty((B, self.L * self.m)); cost = np.empty((B,))
        status = np.empty((B,), dtype=np.int32); iters = np.empty((B,), dtype=np.int32)
        vp = lambda x: C.c_void_p(x.ctypes.data)
        L.check(self._lib.ddmpc_solve_from_host(self._h, *(vp(a) for a in arrs), vp(u_opt), vp(cost), vp(status), vp(iters)))
        self._keep["data"] = (None, None)
        return u_opt, cost, status, iters

    def prepare(self) -> None:
        """Factor once per data set and form the affine law used by `step`."""
        if "data" in self._keep and self._keep["data"][0] is not None and _is_torch(self._keep["data"][0]):
            self._use_torch_stream()
        L.check(self._lib.ddmpc_prepare(self._h))

    def step(self, u_past, y_past, u_opt=None, cost=None, status=None, iters=None):
        """Warm control step (`ddmpc_step`): same outputs as `solve`."""
        return self.solve(u_past, y_past, u_opt, cost, status, iters, warm=True)

    def gain(self) -> np.ndarray:
        """The prepared affine law: [batch, n*(m+p)+1, r] with beta = gain[:,0] + gain[:,1:]^T [u_past; y_past]."""
        nf = self.n * (self.m + self.p)
        out = np.empty((self.batch, nf + 1, (self.m + self.p) * (self.L + self.n)))
        L.check(self._lib.ddmpc_get_gain(self._h, C.c_void_p(out.ctypes.data), L.MEM_HOST))
        return out

    def set_closed_loop_graph(self, on: bool) -> None:
        """Record the per-step launches of `closed_loop` into one HIP graph and replay it (default off)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_CLOSED_LOOP_GRAPH, 1 if on else 0))

    def set_closed_loop_path(self, path: str) -> None:
        """'auto' | 'cold' | 'warm' (DDMPC_OPT_CLOSED_LOOP_PATH)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_CLOSED_LOOP_PATH, {"auto": 0, "cold": 1, "warm": 2}[path]))

    def set_refinement(self, mode: str = "auto", max_passes: Optional[int] = None, res_log10: Optional[float] = None) -> None:
        """Iterative refinement of the cold solve with exact Hankel products (DDMPC_OPT_REFINE): 'off' | 'auto' |
        'always'; `res_log10`: auto mode re-solves an instance with refinement when the relative exact-Hankel residual of
        its plain solve exceeds 10^res_log10 (resolution 0.1; default -10.7)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_REFINE, {"off": L.REFINE_OFF, "auto": L.REFINE_AUTO, "always": L.REFINE_ALWAYS}[mode]))
        if max_passes is not None:
            L.check(self._lib.ddmpc_set_option(self._h, L.OPT_REFINE_MAX, int(max_passes)))
        if res_log10 is not None:
            L.check(self._lib.ddmpc_set_option(self._h, L.OPT_REFINE_RES_LOG10, int(round(-10 * res_log10))))

    def set_large_pipeline(self, mode: str) -> None:
        """NOMINAL controllers beyond the register-resident kernels: 'phases' (default: one kernel per phase over the whole
        batch) | 'one_workgroup' (DDMPC_OPT_LARGE_PIPELINE)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_LARGE_PIPELINE,
                                           {"one_workgroup": L.PIPELINE_ONE_WORKGROUP, "phases": L.PIPELINE_PHASES}[mode]))

    def set_convex_update(self, on: bool) -> None:
        """Slack CONVEX on the register-resident kernels: True (default) = active-set iterations after the first keep the first
        factor (rank-k update), False = every iteration factors the system again (DDMPC_OPT_CONVEX_UPDATE)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_CONVEX_UPDATE, 1 if on else 0))

    def set_gram_launch(self, kind: str) -> None:
        """Structured Gram of plants with other than two or four channels on the register-resident kernels: "matrix_pipe"
        (default: streaming launch, lag sums and window walk by MFMA) or "staged" (round 4's launch with the whole trajectory
        in LDS) -- DDMPC_OPT_GRAM_LAUNCH."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_GRAM_LAUNCH, {"matrix_pipe": 0, "staged": 1}[kind]))

    def set_large_affine_law(self, on: bool) -> None:
        """NOMINAL controllers beyond the register-resident kernels: `prepare` also forms the affine law z(past) and `step`
        evaluates it (DDMPC_OPT_LARGE_AFFINE_LAW; default off: `step` repeats the solve on the kept factors)."""
        L.check(self._lib.ddmpc_set_option(self._h, L.OPT_LARGE_AFFINE_LAW, 1 if on else 0))

    def closed_loop(self, A, B, Cm, D, x0, u_past, y_past, w, n_mpc_step: int = 1):
        """Batched closed loop on the device (controller_operation.py:259-305 for every instance).

        A,B,Cm,D: plant matrices; x0 [batch,ns]; u_past [batch,n*m], y_past [batch,n*p];
        w [batch,n_steps,p] measurement noise.  Returns (u_sys, y_sys, status, x_end, u_past_end, y_past_end)
        as host arrays (inputs are not modified)."""
        A = np.ascontiguousarray(A, dtype=np.float64); Bm = np.ascontiguousarray(B, dtype=np.float64)
        Cm = np.ascontiguousarray(Cm, dtype=np.float64); D = np.ascontiguousarray(D, dtype=np.float64)
        ns = A.shape[0]
        if A.shape != (ns, ns) or Bm.shape != (ns, self.m) or Cm.shape != (self.p, ns) or D.shape != (self.p, self.m):
            raise ValueError("plant matrices have inconsistent shapes")
        w = np.ascontiguousarray(w, dtype=np.float64)
        if w.ndim != 3 or w.shape[0] != self.batch or w.shape[2] != self.p:
            raise ValueError("w must have shape [batch, n_steps, p]")
        n_steps = w.shape[1]
        x = np.array(x0, dtype=np.float64, order="C").reshape(self.batch, ns).copy()
        up = np.array(u_past, dtype=np.float64, order="C").reshape(self.batch, self.n * self.m).copy()
        yp = np.array(y_past, dtype=np.float64, order="C").reshape(self.batch, self.n * self.p).copy()
        u_sys = np.empty((self.batch, n_steps, self.m)); y_sys = np.empty((self.batch, n_steps, self.p))
        status = np.empty((self.batch,), dtype=np.int32)
        pl = L.Plant(ns, A.ctypes.data_as(L.c_double_p), Bm.ctypes.data_as(L.c_double_p),
                     Cm.ctypes.data_as(L.c_double_p), D.ctypes.data_as(L.c_double_p))
        vp = lambda a: C.c_void_p(a.ctypes.data)
        L.check(self._lib.ddmpc_closed_loop(self._h, C.byref(pl), n_steps, int(n_mpc_step), vp(x), vp(up), vp(yp), vp(w),
                                            vp(u_sys), vp(y_sys), vp(status), L.MEM_HOST))
        return u_sys, y_sys, status, x, up, yp

    def persistent_excitation_ranks(self, u_d) -> np.ndarray:
        """Rank of the order-(L+2n) Hankel matrix of every instance's input data, the reference's
        construction-time guard (controller.py:275-296, hankel_matrix.py:55-87): Hankel gather on the
        GPU, ranks by batched SVD on the host with numpy's default tolerance (exactly the reference's
        test).  An instance is persistently exciting iff its rank equals m*(L+2n)."""
        H = hankel_matrix_batched(np.asarray(u_d, dtype=np.float64), self.L + 2 * self.n, device=self.device)
        return np.linalg.matrix_rank(H)

    PE_CERTIFY_RATIO = 1e-5

    def persistent_excitation_guard(self, u_d):
        """Batched construction-time guard (controller.py:275-296): returns (ok [batch] bool, rank [batch] int).

        Device: `ddmpc_pe_guard` gives a rigorous lower bound of sigma_min/sigma_max of every instance's
        order-(L+2n) input Hankel matrix; a bound above PE_CERTIFY_RATIO (1e-5, nine orders above the SVD
        tolerance max(M,N)*eps and three above the Gram's rounding floor) certifies rank m*(L+2n).
        Host: only the instances left undecided get the reference's exact SVD test (numpy default tolerance)."""
        u_d = np.ascontiguousarray(np.asarray(u_d, dtype=np.float64))
        B, N, m = u_d.shape
        order = self.L + 2 * self.n
        full = m * order
        ratio = np.empty((B,))
        L.check(self._lib.ddmpc_pe_guard(C.c_void_p(u_d.ctypes.data), B, N, m, order, C.c_void_p(ratio.ctypes.data),
                                         L.MEM_HOST, self.device))
        rank = np.full((B,), full, dtype=np.int64)
        undecided = np.nonzero(~(ratio > self.PE_CERTIFY_RATIO))[0]
        if undecided.size:
            H = hankel_matrix_batched(u_d[undecided], order, device=self.device)
            rank[undecided] = np.linalg.matrix_rank(H)
        return rank == full, rank

    def set_setpoints(self, u_s, y_s) -> None:
        us = np.ascontiguousarray(np.asarray(u_s, dtype=np.float64).reshape(-1))
        ys = np.ascontiguousarray(np.asarray(y_s, dtype=np.float64).reshape(-1))
        if us.size != self.m or ys.size != self.p:
            raise ValueError("u_s / y_s must have m / p entries")
        L.check(self._lib.ddmpc_set_setpoints(self._h, C.c_void_p(us.ctypes.data), C.c_void_p(ys.ctypes.data)))
        self._us, self._ys = us, ys

    def get_solution(self, what: str) -> np.ndarray:
        """`.value` of alpha / ubar / ybar / sigma after the last solve (host array)."""
        sel = {"alpha": L.SOL_ALPHA, "ubar": L.SOL_UBAR, "ybar": L.SOL_YBAR, "sigma": L.SOL_SIGMA}[what]
        Ln = self.L + self.n
        per = {"alpha": self.N - Ln + 1, "ubar": Ln * self.m, "ybar": Ln * self.p, "sigma": Ln * self.p}[what]
        out = np.empty((self.batch, per))
        L.check(self._lib.ddmpc_get_solution(self._h, sel, C.c_void_p(out.ctypes.data), L.MEM_HOST))
        return out

    def synchronize(self) -> None:
        L.check(self._lib.ddmpc_synchronize(self._h))

    def cost_model(self) -> Tuple[float, float]:
        f, b = C.c_double(), C.c_double()
        L.check(self._lib.ddmpc_cost_model(self._h, C.byref(f), C.byref(b)))
        return f.value, b.value

    def debug_stamps(self, enable: bool = True, fetch: bool = False):
        """Diagnostics: enable in-kernel phase stamps / fetch those of the last solve ([batch,16] uint64)."""
        out = np.zeros((self.batch, 16), dtype=np.uint64) if fetch else None
        L.check(self._lib.ddmpc_debug_stamps(self._h, 1 if enable else 0,
                                             C.c_void_p(out.ctypes.data) if fetch else C.c_void_p()))
        return out

    def kernel_name(self) -> str:
        return self._lib.ddmpc_kernel_name(self._h).decode()


def hankel_matrix_batched(X: np.ndarray, L_: int, device: int = 0) -> np.ndarray:
    """Batched hankel_matrix on the GPU: X [B,N,nch] -> [B, L*nch, N-L+1]."""
    lib = L.load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    B, N, nch = X.shape
    if N < L_:
        raise ValueError("N must be greater than or equal to L.")
    H = np.empty((B, L_ * nch, N - L_ + 1))
    L.check(lib.ddmpc_hankel(C.c_void_p(X.ctypes.data), B, N, nch, L_, C.c_void_p(H.ctypes.data), L.MEM_HOST, device))
    return H
