"""Multi-restart L-BFGS-B driver — counterpart of BOBE/optim.py:249-359 (``optimize_scipy``).

The three drivers keep the reference's call signature ``(fun, fun_args=(), fun_kwargs={}, num_params=1, bounds=None,
x0=None, optimizer_options=..., maxiter=200, n_restarts=..., verbose=False)`` (optim.py:18-28, 166-177, 249-260;
pinned by tests/golden/reference_signatures.json).  The reference wraps ``jax.jit(jax.value_and_grad(fun))``
(optim.py:306-309); there is no autodiff here, so ``fun(x, *fun_args, **fun_kwargs)`` hands back ``(f, g)`` itself - the
heavy part running on the GPU through the C ABI - and a ``fun`` that returns the value alone is differentiated by
forward differences.  What this build adds (batched / slotted evaluation of the restarts) is keyword-only.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Tuple

import numpy as np
from scipy.optimize import minimize

from .utils import get_logger

log = get_logger("optim")


def _bind(fun: Callable, fun_args, fun_kwargs) -> Callable:
    """x -> (f, g | None) from the reference's ``fun, fun_args, fun_kwargs`` triple: ``fun`` returns ``(f, g)`` (what
    ``jax.value_and_grad(fun)`` returns there, optim.py:306-309) or the value alone (g = None: finite differences)."""
    args = tuple(fun_args) if fun_args else ()
    kwargs = dict(fun_kwargs) if fun_kwargs else {}

    def vg(x):
        r = fun(x, *args, **kwargs)
        if isinstance(r, (tuple, list)) and len(r) == 2:
            return r[0], r[1]
        return r, None
    return vg


def _with_fd_grad(vg: Callable) -> Callable:
    """(f, g) for a value-only objective: forward differences, the step SciPy's '2-point' scheme takes."""
    from scipy.optimize import approx_fprime

    def full(x):
        x = np.asarray(x, dtype=np.float64)
        return float(vg(x)[0]), approx_fprime(x, lambda q: float(vg(q)[0]), np.sqrt(np.finfo(float).eps))
    return full


def _setup_bounds(bounds, num_params):
    """BOBE/optim.py:42-68."""
    if bounds is None:
        return None
    bounds = np.array(bounds, dtype=np.float64)
    if bounds.shape == (2,):
        bounds = np.tile(bounds.reshape(1, 2), (num_params, 1)).T
    elif bounds.shape != (2, num_params):
        raise ValueError(f"Bounds shape {bounds.shape} incompatible with {num_params} parameters")
    return bounds


class _LockStep:
    """Runs R independent SciPy minimisations in R threads and serves their objective calls in rounds: when
    every still-running minimisation has asked for a value, the pending points go to ``batch_fun`` in ONE call
    (restart order), so the GPU evaluates them concurrently.  Each minimisation sees exactly the values it would
    have seen alone; the rounds are deterministic, and ``batch_fun`` is only ever called from one thread at a time.
    """

    def __init__(self, batch_fun: Callable, n_workers: int):
        self.batch_fun = batch_fun
        self.cv = threading.Condition()
        self.active = n_workers
        self.pending = {}          # worker -> x
        self.results = {}          # worker -> (f, g) or exception

    def _flush_locked(self):
        ids = sorted(self.pending)
        xs = [self.pending[i] for i in ids]
        self.pending = {}
        try:
            out = self.batch_fun(xs)
            for i, r in zip(ids, out):
                self.results[i] = r
        except Exception:               # narrowed down point by point: only the minimisation whose point raises gets it
            for i, x in zip(ids, xs):
                try:
                    self.results[i] = self.batch_fun([x])[0]
                except Exception as e:
                    self.results[i] = e
        self.cv.notify_all()

    def call(self, wid: int, x):
        with self.cv:
            self.pending[wid] = np.array(x, dtype=np.float64)
            if len(self.pending) == self.active:
                self._flush_locked()
            while wid not in self.results:
                self.cv.wait()
            r = self.results.pop(wid)
        if isinstance(r, Exception):
            raise r
        return r

    def done(self, wid: int):
        with self.cv:
            self.active -= 1
            if self.active > 0 and len(self.pending) == self.active:
                self._flush_locked()


def _minimize_concurrently(batch_fun: Callable, starts: np.ndarray, has_grad: bool, **kw) -> List:
    """``scipy.optimize.minimize`` from every row of ``starts``; returns the results (or the exception) per row.
    L-BFGS-B with gradients is stepped by one thread (``_rc_minimize_all``) where that is available."""
    known = {"maxcor", "ftol", "gtol", "maxfun", "maxiter", "maxls"}
    if has_grad and kw.get("method", "L-BFGS-B") == "L-BFGS-B" and set(kw.get("options") or {}) <= known and _rc_available():
        try:
            return _rc_minimize_all(batch_fun, starts, kw.get("bounds"), dict(kw.get("options") or {}))
        except Exception as e:  # pragma: no cover
            log.warning(f"stepped L-BFGS-B driver failed with {e}; running the restarts in threads")
    ls = _LockStep(batch_fun, len(starts))
    out: List = [None] * len(starts)

    def work(i):
        try:
            f = (lambda x: ls.call(i, x)) if has_grad else (lambda x: ls.call(i, x)[0])
            out[i] = minimize(f, starts[i], jac=has_grad, **kw)
        except Exception as e:
            out[i] = e
        finally:
            ls.done(i)

    threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(starts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out


# --------------------------------------------------------------------------------------------------------------
# The same lock step without threads: SciPy's L-BFGS-B is a reverse-communication routine (``setulb`` hands control back
# whenever it wants f and g), so R minimisations can be advanced by ONE thread — every routine is stepped until it asks
# for a value, the pending points go to ``batch_fun`` in one call, the values are handed back.  No condition variables, no
# GIL hand-offs between R threads: at the sizes of the BO loop (N = 100 ... 1200), where an evaluation is a ~100 us
# graph replay, those cost more than the evaluation itself.  This is the loop of scipy.optimize._lbfgsb_py
# ._minimize_lbfgsb (as of SciPy 1.15), statement for statement, per restart; it uses a private SciPy module, so the
# routine's argument list is checked (``_setulb_mismatch``) and the loop is verified against ``minimize`` on a quadratic at
# first use; the thread driver above stays as the fallback.
# --------------------------------------------------------------------------------------------------------------
_RC_STATE = {"checked": False, "ok": False}


class _RcRun:
    """One L-BFGS-B minimisation stepped from outside (the body of ``_minimize_lbfgsb`` up to its ``func_and_grad`` call)."""

    def __init__(self, lbfgsb, x0, bounds, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000,
                 maxls=20, **unknown):
        self.lb = lbfgsb
        self.m, self.maxiter, self.maxfun, self.maxls = int(maxcor), int(maxiter), int(maxfun), int(maxls)
        self.pgtol = gtol
        self.factr = ftol / np.finfo(float).eps
        x0 = np.asarray(x0, dtype=np.float64).ravel()
        n = x0.shape[0]
        self.nbd = np.zeros(n, np.int32)
        self.low = np.zeros(n, np.float64)
        self.up = np.zeros(n, np.float64)
        if bounds is not None:
            lo = np.array([(-np.inf if b[0] is None else b[0]) for b in bounds], dtype=np.float64)
            hi = np.array([(np.inf if b[1] is None else b[1]) for b in bounds], dtype=np.float64)
            if (lo > hi).any():
                raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
            x0 = np.clip(x0, lo, hi)
            bounds_map = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
            for i in range(n):
                fl, fu = bool(np.isfinite(lo[i])), bool(np.isfinite(hi[i]))
                if fl:
                    self.low[i] = lo[i]
                if fu:
                    self.up[i] = hi[i]
                self.nbd[i] = bounds_map[fl, fu]
        m = self.m
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)
        self.g = np.zeros((n,), dtype=np.int32)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.nit, self.nfev = 0, 0
        self.finished = False
        self.last_x, self.last_f, self.last_g = None, None, None     # ScalarFunction's one-point cache

    def advance(self):
        """Step until the routine wants f and g at a NEW point ``self.x`` (returns True) or has finished (False).  A
        request at the point evaluated last (the routine restarts an iteration there after a failed line search) is
        answered from the cache, as ScalarFunction.fun_and_grad does: no evaluation, not counted."""
        while True:
            self.g = self.g.astype(np.float64)
            self.lb.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa,
                           self.iwa, self.task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if self.task[0] == 3:
                if self.last_x is not None and np.array_equal(self.x, self.last_x):
                    self.f, self.g = self.last_f, self.last_g
                    continue
                return True
            if self.task[0] == 1:                       # new iteration
                self.nit += 1
                if self.nit >= self.maxiter:
                    self.task[0], self.task[1] = 5, 504
                elif self.nfev > self.maxfun:
                    self.task[0], self.task[1] = 5, 502
            else:
                self.finished = True
                return False

    def give(self, f, g):
        self.nfev += 1
        self.f = f
        self.g = np.asarray(g, dtype=np.float64)
        self.last_x, self.last_f, self.last_g = np.array(self.x), self.f, self.g

    def result(self):
        from scipy.optimize import OptimizeResult
        from scipy.optimize._lbfgsb_py import status_messages, task_messages
        if self.task[0] == 4:
            warnflag = 0
        elif self.nfev > self.maxfun or self.nit >= self.maxiter:
            warnflag = 1
        else:
            warnflag = 2
        msg = status_messages[self.task[0]] + ": " + task_messages[self.task[1]]
        return OptimizeResult(fun=self.f, jac=self.g, nfev=self.nfev, njev=self.nfev, nit=self.nit, status=warnflag,
                              message=msg, x=self.x, success=(warnflag == 0))


def _rc_minimize_all(batch_fun: Callable, starts, bounds, options) -> List:
    """Every restart stepped by this thread; returns an OptimizeResult - or the exception that ended it - per restart.
    An exception out of ``batch_fun`` is narrowed down by evaluating that round's points one by one: only the restart
    whose point raises ends with it (what ``_LockStep`` / one ``minimize`` per thread would do), the others go on."""
    from scipy.optimize import _lbfgsb
    runs = [_RcRun(_lbfgsb, x0, bounds, **options) for x0 in starts]
    failed: dict = {}
    waiting = [i for i, r in enumerate(runs) if r.advance()]
    while waiting:
        # (ScalarFunction hands the objective a copy of x and stores float(f): the values the routine sees are the same)
        pts = [np.array(runs[i].x) for i in waiting]
        try:
            vals = batch_fun(pts)
        except Exception:
            vals = []
            for x in pts:
                try:
                    vals.append(batch_fun([x])[0])
                except Exception as e:
                    vals.append(e)
        nxt = []
        for i, v in zip(waiting, vals):
            if isinstance(v, Exception):
                failed[i] = v
                continue
            f, g = v
            runs[i].give(float(f), g)
            if runs[i].advance():
                nxt.append(i)
        waiting = nxt
    return [failed[i] if i in failed else r.result() for i, r in enumerate(runs)]


def lbfgs_driver() -> str:
    """Which driver ``optimize_scipy`` uses for concurrent L-BFGS-B restarts with a batch objective: "stepped" (SciPy's
    reverse-communication routine stepped by one thread, ``_rc_minimize_all``) or "threads" (one ``minimize`` per
    restart in lock step, ``_LockStep``)."""
    return "stepped" if _rc_available() else "threads"


_SETULB_ARGS = ("m", "x", "l", "u", "nbd", "f", "g", "factr", "pgtol", "wa", "iwa", "task", "lsave", "isave", "dsave", "maxls",
                "ln_task")


def _setulb_mismatch() -> Optional[str]:
    """None when SciPy's reverse-communication routine takes the arguments ``_RcRun.advance`` hands it, else the reason.
    What is tested is the routine itself - its name, its argument list as its docstring states it, the two message
    tables ``_RcRun.result`` reads - not SciPy's version text: any release that keeps them takes the stepped driver (and
    still has to reproduce ``minimize`` bit for bit in the self-check below)."""
    try:
        from scipy.optimize import _lbfgsb, _lbfgsb_py
    except Exception as e:
        return f"scipy.optimize._lbfgsb is not importable ({e})"
    fn = getattr(_lbfgsb, "setulb", None)
    if fn is None:
        return "scipy.optimize._lbfgsb has no setulb"
    doc = (getattr(fn, "__doc__", None) or "").strip().splitlines()
    head = doc[0].replace(" ", "") if doc else ""
    # "setulb(m,x,...,ln_task)" (the C routine of SciPy >= 1.15) - an f2py wrapper writes "... = setulb(m,x,...,[n,...])"
    if "setulb(" not in head or ")" not in head:
        return "setulb states no argument list"
    args = head[head.index("setulb(") + len("setulb("):head.rindex(")")]
    required = tuple(a for a in args.split("[")[0].split(",") if a)
    if required != _SETULB_ARGS:
        return f"setulb takes ({', '.join(required)}), the stepped driver was written for ({', '.join(_SETULB_ARGS)})"
    if not (hasattr(_lbfgsb_py, "status_messages") and hasattr(_lbfgsb_py, "task_messages")):
        return "scipy.optimize._lbfgsb_py has no status_messages / task_messages"
    return None


def _rc_available() -> bool:
    """The private routine is there, takes the arguments this file hands it (``_setulb_mismatch``), and the stepped loop
    reproduces ``minimize`` exactly on a bounded quadratic (checked once per process)."""
    if _RC_STATE["checked"]:
        return _RC_STATE["ok"]
    _RC_STATE["checked"] = True
    try:
        import scipy
        why = _setulb_mismatch()
        if why:
            log.warning(f"SciPy {scipy.__version__}: {why}; the stepped L-BFGS-B driver is switched off and concurrent "
                        "restarts run one `minimize` per thread (same results, slower at BO-loop sizes)")
            return False
        A = np.array([[3.0, 0.4, 0.1], [0.4, 2.0, -0.3], [0.1, -0.3, 1.5]])
        b = np.array([1.0, -2.0, 0.5])

        def vg(x):
            return float(0.5 * x @ A @ x - b @ x + 0.1 * np.sum(x ** 4)), A @ x - b + 0.4 * x ** 3
        bounds = [(-0.5, 2.0), (-3.0, 0.2), (None, None)]
        starts = np.array([[1.5, -2.5, 3.0], [-0.2, 0.1, -4.0]])
        opts = {"ftol": 1e-9, "gtol": 1e-8, "maxiter": 50}
        mine = _rc_minimize_all(lambda xs: [vg(x) for x in xs], starts, bounds, opts)
        for x0, r in zip(starts, mine):
            ref = minimize(vg, x0, jac=True, method="L-BFGS-B", bounds=bounds, options=dict(opts))
            if not (np.array_equal(r.x, ref.x) and r.fun == ref.fun and r.nit == ref.nit and r.nfev == ref.nfev and
                    r.message == ref.message and bool(r.success) == bool(ref.success)):
                log.warning("the stepped L-BFGS-B driver does not reproduce scipy.optimize.minimize on this SciPy build; "
                            "concurrent restarts run one `minimize` per thread")
                return False
        _RC_STATE["ok"] = True
        log.info(f"concurrent L-BFGS-B restarts: stepped driver (SciPy {scipy.__version__}, verified against minimize)")
    except Exception as e:  # pragma: no cover
        log.warning(f"stepped L-BFGS-B driver unavailable ({e}); concurrent restarts run one `minimize` per thread")
    return _RC_STATE["ok"]


def _minimize_in_slots(slot_fun: Callable, starts: np.ndarray, has_grad: bool, n_slots: int, **kw) -> List:
    """``scipy.optimize.minimize`` from every row of ``starts`` on ``n_slots`` evaluation slots: one worker thread
    per slot, worker w running the restarts w, w + n_slots, ... one after the other through ``slot_fun(x, w)``.
    Nothing synchronises the workers: each advances as fast as its own evaluations return.  (Never more threads
    than slots: threads queueing for a shared slot were measured 20x slower than the sequential loop.)"""
    out: List = [None] * len(starts)
    n_workers = max(1, min(n_slots, len(starts)))

    def work(slot):
        def f(x):
            r = slot_fun(x, slot)
            return r if has_grad else r[0]
        for i in range(slot, len(starts), n_workers):
            try:
                out[i] = minimize(f, starts[i], jac=has_grad, **kw)
            except Exception as e:
                out[i] = e

    threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(n_workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out


def _evaluate_in_slots(slot_fun: Callable, xs, n_slots: int) -> List:
    """slot_fun(x, slot) for every x, up to n_slots at a time (one thread per slot)."""
    xs = list(xs)
    out: List = [None] * len(xs)

    def work(slot):
        for i in range(slot, len(xs), n_slots):
            out[i] = slot_fun(xs[i], slot)

    threads = [threading.Thread(target=work, args=(s_,), daemon=True) for s_ in range(min(n_slots, len(xs)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out


def optimize_scipy(fun: Callable, fun_args=(), fun_kwargs=None, num_params: int = 1, bounds=None, x0=None,
                   optimizer_options: Optional[dict] = None, maxiter: int = 200, n_restarts: int = 4,
                   verbose: bool = False, *, batch_value_and_grad: Optional[Callable] = None,
                   slot_value_and_grad: Optional[Callable] = None, n_slots: int = 4) -> Tuple[np.ndarray, float]:
    """Same arguments and the same restart / screening / acceptance logic as BOBE/optim.py:249-359.

    ``fun(x, *fun_args, **fun_kwargs) -> (f, g)``; with ``g`` None or a bare value SciPy falls back to finite differences.
    ``batch_value_and_grad(list of x) -> list of (f, g)`` (optional) lets the restarts — independent L-BFGS-B
    runs that the reference walks one after the other — advance in lock-step with their objective evaluations
    batched on the GPU; every restart follows the same trajectory and the acceptance order is unchanged.
    ``slot_value_and_grad(x, slot) -> (f, g)`` (optional, takes precedence) runs every restart in its own thread on
    one of ``n_slots`` evaluation slots without any barrier between them; same trajectories, same acceptance order.
    """
    options = dict(optimizer_options if optimizer_options is not None else
                   {"method": "L-BFGS-B", "ftol": 1e-6, "gtol": 1e-6})      # optim.py:256
    options.update({"maxiter": maxiter})                                   # optim.py:292
    method = options.pop("method", "L-BFGS-B")                             # optim.py:294
    bounds_arr = _setup_bounds(bounds, num_params)
    scipy_bounds = None if bounds_arr is None else [
        (float(bounds_arr[0, i]), float(bounds_arr[1, i])) for i in range(num_params)]
    if x0 is None:
        raise ValueError("x0 must be provided (shape: (n_restarts, num_params) or (num_params,))")
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    if x0.shape[0] < n_restarts:
        raise ValueError(f"x0 provided with {x0.shape[0]} restarts but n_restarts={n_restarts}")
    x0 = x0[:n_restarts]

    value_and_grad = _bind(fun, fun_args, fun_kwargs)
    probe = value_and_grad(x0[0])
    has_grad = probe[1] is not None
    objective = value_and_grad if has_grad else (lambda x: value_and_grad(x)[0])

    slotted = slot_value_and_grad is not None and len(x0) > 1
    concurrent = batch_value_and_grad is not None and len(x0) > 1 and not slotted
    best_f, best_x = np.inf, None
    screen = None
    if slotted:
        try:
            screen = [probe] + _evaluate_in_slots(slot_value_and_grad, x0[1:], n_slots)
        except Exception as e:  # pragma: no cover
            log.warning(f"  concurrent screening failed with {e}; evaluating the starts one by one")
    if concurrent:
        try:
            screen = [probe] + list(batch_value_and_grad([x for x in x0[1:]]))
        except Exception as e:  # pragma: no cover
            log.warning(f"  batched screening failed with {e}; evaluating the starts one by one")
    for i, x_init in enumerate(x0):                                        # optim.py:325-333
        try:
            val = screen[i][0] if screen is not None else (probe[0] if i == 0 else value_and_grad(x_init)[0])
            if np.isfinite(val) and val < best_f:
                best_f, best_x = float(val), np.array(x_init)
        except Exception as e:  # pragma: no cover
            log.warning(f"  Initial point {i + 1}/{n_restarts}: failed with {e}")
    results = None
    if slotted:
        results = _minimize_in_slots(slot_value_and_grad, x0, has_grad, n_slots, method=method, bounds=scipy_bounds,
                                     options=options)
    elif concurrent:
        results = _minimize_concurrently(batch_value_and_grad, x0, has_grad, method=method, bounds=scipy_bounds,
                                         options=options)
    for i, x_init in enumerate(x0):                                        # optim.py:335-354
        try:
            if results is not None:
                res = results[i]
                if isinstance(res, Exception):
                    raise res
            else:
                res = minimize(objective, x_init, method=method, jac=has_grad, bounds=scipy_bounds, options=options)
        except Exception as e:
            if verbose:
                log.warning(f"  Restart {i + 1}/{n_restarts}: failed with {e}")
            continue
        ok = res.success or "ITERATIONS REACHED LIMIT" in str(res.message).upper()   # optim.py:340
        if ok and np.isfinite(res.fun) and res.fun < best_f:
            best_f, best_x = float(res.fun), np.array(res.x)
    if best_x is None:
        best_x = np.array(x0[0])
    return np.asarray(best_x), float(best_f)


# --------------------------------------------------------------------------------------------------------------
# first-order optimisers (BOBE/optim.py:18-247: optimize_optax / optimize_optax_vmap)
# --------------------------------------------------------------------------------------------------------------
class _FirstOrder:
    """The two optax transformations the reference names explicitly (optim.py:30-33), restated: ``adam`` (b1 = 0.9,
    b2 = 0.999, eps = 1e-8, bias-corrected, update -lr * m_hat / (sqrt(v_hat) + eps)) and ``sgd`` (optional
    ``momentum``).  State arrays carry a leading restart axis so that the vectorised driver updates all restarts at
    once."""

    def __init__(self, name: str, learning_rate: float, **kw):
        self.name = name.lower()
        if self.name not in ("adam", "sgd"):
            raise ValueError(f"Optimizer '{name}' is not available (adam and sgd are built)")
        self.lr = float(kw.pop("learning_rate", learning_rate))
        self.b1, self.b2, self.eps = float(kw.pop("b1", 0.9)), float(kw.pop("b2", 0.999)), float(kw.pop("eps", 1e-8))
        self.momentum = kw.pop("momentum", None)
        if kw:
            raise ValueError(f"unsupported optimizer options: {sorted(kw)}")

    def init(self, params):
        z = np.zeros_like(params)
        return {"t": 0, "m": z.copy(), "v": z.copy()}

    def update(self, grad, state):
        state["t"] += 1
        if self.name == "sgd":
            if self.momentum:
                state["m"] = self.momentum * state["m"] + grad
                return -self.lr * state["m"]
            return -self.lr * grad
        state["m"] = self.b1 * state["m"] + (1.0 - self.b1) * grad
        state["v"] = self.b2 * state["v"] + (1.0 - self.b2) * grad * grad
        m_hat = state["m"] / (1.0 - self.b1 ** state["t"])
        v_hat = state["v"] / (1.0 - self.b2 ** state["t"])
        return -self.lr * m_hat / (np.sqrt(v_hat) + self.eps)


def _first_order_setup(optimizer_options, num_params, bounds, x0, n_restarts, exact_restarts):
    options = dict(optimizer_options if optimizer_options is not None else {})
    patience = options.pop("early_stop_patience", 25)                      # optim.py:104-106
    lr = options.pop("lr", 1e-3)
    opt = _FirstOrder(options.pop("name", "adam"), lr, **options)
    if x0 is None:
        raise ValueError("x0 must be provided (shape: (n_restarts, num_params))")
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    if (x0.shape[0] != n_restarts) if exact_restarts else (x0.shape[0] < n_restarts):
        raise ValueError(f"x0 provided with {x0.shape[0]} restarts but n_restarts={n_restarts}")
    bounds_arr = _setup_bounds(bounds, num_params)
    if bounds_arr is not None:
        span = bounds_arr[1] - bounds_arr[0]

        def to_x(u):
            return u * span + bounds_arr[0]                                # scale_from_unit
    else:
        span = None

        def to_x(u):
            return u
    return opt, patience, x0[:n_restarts], bounds_arr, span, to_x


def optimize_optax(fun: Callable, fun_args=(), fun_kwargs=None, num_params: int = 1, bounds=None, x0=None,
                   optimizer_options: Optional[dict] = None, maxiter: int = 200, n_restarts: int = 1,
                   verbose: bool = False) -> Tuple[np.ndarray, float]:
    """Restart-after-restart first-order minimisation, the loop of BOBE/optim.py:71-163 with ``fun(x, *fun_args,
    **fun_kwargs) -> (f, g)`` in place of ``jax.value_and_grad(fun)``: the iterate lives in unit coordinates of ``bounds`` and is
    clipped to [0, 1] after every step; a restart stops after ``early_stop_patience`` steps without a new best value;
    the value reported for a restart is the best value seen, the point its LAST iterate (optim.py:156-158).
    As in the reference, the rows of ``x0`` are taken as they are as the first iterates (optim.py:138)."""
    opt, patience0, x0, bounds_arr, span, to_x = _first_order_setup(optimizer_options, num_params, bounds, x0, n_restarts, False)
    value_and_grad = _bind(fun, fun_args, fun_kwargs)
    if value_and_grad(to_x(x0[0]))[1] is None:
        value_and_grad = _with_fd_grad(value_and_grad)

    def vg_unit(u):
        f, g = value_and_grad(to_x(u))
        return float(f), (np.asarray(g, dtype=np.float64) * span if span is not None else np.asarray(g, dtype=np.float64))

    best_f, best_u = np.inf, None
    for x_init in x0:                                                      # optim.py:124-132
        try:
            val = vg_unit(x_init)[0]
            if np.isfinite(val) and val < best_f:
                best_f, best_u = val, np.array(x_init)
        except Exception as e:  # pragma: no cover
            log.warning(f"  Initial point: failed with {e}")
    for x_init in x0:                                                      # optim.py:134-160
        u = np.array(x_init)
        state = opt.init(u)
        best_restart, patience = np.inf, patience0
        for _ in range(maxiter):
            f, g = vg_unit(u)
            u = u + opt.update(g, state)
            if bounds_arr is not None:
                u = np.clip(u, 0.0, 1.0)
            if f < best_restart:
                best_restart, patience = f, patience0
            else:
                patience -= 1
                if patience == 0:
                    break
        if best_restart < best_f:
            best_f, best_u = best_restart, u
    if best_u is None:
        best_u = np.array(x0[0])
    return np.asarray(to_x(best_u)), float(best_f)


def optimize_optax_vmap(fun: Callable, fun_args=(), fun_kwargs=None, num_params: int = 1, bounds=None, x0=None,
                        optimizer_options: Optional[dict] = None, maxiter: int = 200, n_restarts: int = 1,
                        verbose: bool = False, *, batch_value_and_grad: Optional[Callable] = None
                        ) -> Tuple[np.ndarray, float]:
    """All restarts step together (BOBE/optim.py:166-247, ``jax.vmap`` of ``fun`` over the restarts): one call of
    ``batch_value_and_grad(list of x) -> list of (f, g)`` per iteration — on this engine one ``bobe_gp_mll_batch``
    with every restart's evaluation in flight (without it: ``fun(x, *fun_args, **fun_kwargs) -> (f, g)`` restart by
    restart).  Per-restart best value / best iterate bookkeeping and the joint early stop of optim.py:228-236."""
    opt, patience0, x0, bounds_arr, span, to_x = _first_order_setup(optimizer_options, num_params, bounds, x0, n_restarts, True)
    if batch_value_and_grad is None:
        single = _bind(fun, fun_args, fun_kwargs)
        if single(to_x(x0[0]))[1] is None:
            single = _with_fd_grad(single)

        def batch_value_and_grad(xs):
            return [single(x) for x in xs]
    U = np.array(x0)
    state = opt.init(U)
    best_vals = np.full(n_restarts, np.inf)
    best_params = np.zeros_like(U)
    patience = np.full(n_restarts, patience0, dtype=np.int64)
    for _ in range(maxiter):
        out = batch_value_and_grad([to_x(u) for u in U])
        vals = np.array([float(o[0]) for o in out])
        G = np.array([np.asarray(o[1], dtype=np.float64) for o in out])
        if span is not None:
            G = G * span
        U = U + opt.update(G, state)
        if bounds_arr is not None:
            U = np.clip(U, 0.0, 1.0)
        improved = vals < best_vals
        best_vals = np.where(improved, vals, best_vals)
        best_params = np.where(improved[:, None], U, best_params)
        patience = np.where(improved, patience0, patience - 1)
        if np.all(patience <= 0):
            break
    i = int(np.argmin(best_vals))
    return np.asarray(to_x(best_params[i])), float(best_vals[i])
