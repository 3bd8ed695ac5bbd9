"""How well does the solve kernel pack its robots onto the 1024 wave slots (256 CUs x 4 SIMDs, one wavefront per robot)?
Per-robot cycles of the last solve (profile slot 15, always filled) vs the kernel's HIP-event time:
  lower bound  = sum of cycles / slots;   list schedule = greedy in dispatch order (longest first by the PREVIOUS solve's cycles,
  what order_block does) and in the ideal order (longest first by this solve's own cycles).
usage: python tools/tail_model.py [n_robots] [shader GHz]"""
import heapq, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.09
slots = 1024
dev = torch.device("cuda:0")
wl = make_solver_workload(n, h=10, seed=1000, config=2)
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
solver = BatchedConvexMpc(wl.mass, inertia9, 10, wl.dt_mpc, wl.alpha, device=dev)
solver.enable_timing()


def schedule(order, cyc):
    free = [0] * slots
    heapq.heapify(free)
    end = 0
    for r in order:
        t = heapq.heappop(free) + cyc[r]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


prev = None
w = wl
for s in range(8):
    solver.solve(torch.from_numpy(w.inputs).to(dev))
    torch.cuda.synchronize()
    cyc = np.abs(solver.get_profile()[:, 15]).astype(np.int64)
    _, c = solver.kernel_times(1)
    if prev is not None:
        lb = cyc.sum() / slots
        by_prev = schedule(np.argsort(-prev, kind="stable"), cyc)
        ideal = schedule(np.argsort(-cyc, kind="stable"), cyc)
        fifo = schedule(np.arange(n), cyc)
        print(f"step {s}: kernel {c[-1]:.4f} ms | mean {cyc.mean() / 1e3:.0f} k max {cyc.max() / 1e3:.0f} k cycles | at {ghz} GHz: lower bound {lb / ghz / 1e6:.4f} ms, "
              f"longest-first by previous solve {by_prev / ghz / 1e6:.4f}, by own cycles {ideal / ghz / 1e6:.4f}, robot order {fifo / ghz / 1e6:.4f}")
    prev = cyc
    w = perturb_workload(w, 7000 + 131 * s)

# ---- what predicts a robot's cycles?  (previous cycles; did its contact table change?) ----
if "--predict" in sys.argv:
    from rl_mpc_locomotion_amd import layout as L
    w = wl
    prev_c, prev_tab, prev_it = None, None, None
    for s in range(10):
        f, info = solver.solve(torch.from_numpy(w.inputs).to(dev))
        torch.cuda.synchronize()
        cyc = np.abs(solver.get_profile()[:, 15]).astype(np.float64)
        it = info.cpu().numpy()[:, 0]
        tab = w.inputs[:, L.IN_CONTACT:L.IN_CONTACT + 40].copy()
        if prev_c is not None:
            changed = (tab[:, :4] != prev_tab[:, :4]).any(1)        # stance set of the first horizon step changed
            ratio = cyc / prev_c
            big = ratio > 1.4
            print(f"step {s}: first-row contact changed on {changed.mean():.3f} of robots; cycles grew > 1.4x on {big.mean():.4f}; "
                  f"P(grew | changed) {big[changed].mean() if changed.any() else 0:.3f}  P(grew | same) {big[~changed].mean():.4f}; "
                  f"iters prev->now of the 5 largest: {[(int(prev_it[i]), int(it[i])) for i in np.argsort(-cyc)[:5]]}")
        prev_c, prev_tab, prev_it = cyc, tab, it
        w = perturb_workload(w, 7000 + 131 * s)

# ---- alternative predictors for the dispatch order, replayed through the list scheduler ----
if "--orders" in sys.argv:
    w = wl
    hist = []
    res = {}
    for s in range(34):
        solver.solve(torch.from_numpy(w.inputs).to(dev))
        torch.cuda.synchronize()
        cyc = np.abs(solver.get_profile()[:, 15]).astype(np.float64)
        if len(hist) >= 10:
            preds = {"previous": hist[-1], "max of last 2": np.maximum(hist[-1], hist[-2]), "max of last 3": np.maximum.reduce(hist[-3:]),
                     "one gait period ago (10)": hist[-10], "max(previous, 10 ago)": np.maximum(hist[-1], hist[-10]),
                     "mean of last 4": np.mean(hist[-4:], axis=0), "max of last 10": np.maximum.reduce(hist[-10:]), "clairvoyant": cyc}
            for k, p in preds.items():
                res.setdefault(k, []).append(schedule(np.argsort(-p, kind="stable"), cyc) / ghz / 1e6)
        hist.append(cyc)
        w = perturb_workload(w, 7000 + 131 * s)
    for k, v in res.items():
        print(f"order by {k:28s}: mean {np.mean(v):.4f} ms  max {np.max(v):.4f}  (over {len(v)} steps)")

# ---- what would a separate polish launch buy?  (ADMM part: measured cycles minus a constant polish; polish: uniform jobs) ----
if "--split" in sys.argv:
    w = wl
    polish, reload = 80e3, 8e3
    prev = None
    for s in range(10):
        solver.solve(torch.from_numpy(w.inputs).to(dev))
        torch.cuda.synchronize()
        cyc = np.abs(solver.get_profile()[:, 15]).astype(np.float64)
        if prev is not None and s >= 4:
            one = schedule(np.argsort(-prev, kind="stable"), cyc)
            a = schedule(np.argsort(-prev, kind="stable"), cyc - polish)
            b = int(np.ceil(n / slots)) * (polish + reload)
            print(f"step {s}: one launch {one / ghz / 1e6:.4f} ms | ADMM launch {a / ghz / 1e6:.4f} + polish launch {b / ghz / 1e6:.4f} + ~0.008 gap = {(a + b) / ghz / 1e6 + 0.008:.4f} ms")
        prev = cyc
        w = perturb_workload(w, 7000 + 131 * s)

# ---- would a better static packing of the PREDICTED lengths, replayed greedily with the ACTUAL ones, beat longest-first? ----
if "--packing" in sys.argv:
    def multifit_order(pred):
        """FFD inside a binary search on the bin capacity (multifit); returns the dispatch order = jobs sorted by planned start time."""
        idx = np.argsort(-pred, kind="stable")
        lo, hi = max(pred.sum() / slots, pred.max()), 2 * max(pred.sum() / slots, pred.max())
        best = None
        for _ in range(12):
            cap = 0.5 * (lo + hi)
            load = np.zeros(slots); assign = [[] for _ in range(slots)]
            ok = True
            for j in idx:
                fits = np.nonzero(load + pred[j] <= cap)[0]
                if len(fits) == 0: ok = False; break
                b = fits[0]; assign[b].append(j); load[b] += pred[j]
            if ok: hi, best = cap, assign
            else: lo = cap
        starts = []
        for b in best:
            t = 0.0
            for j in b: starts.append((t, -pred[j], j)); t += pred[j]
        starts.sort()
        return np.array([j for _, _, j in starts]), hi
    w = wl
    hist = []
    res = {"longest-first by max of last 10": [], "multifit on max of last 10": [], "longest-first clairvoyant": [], "multifit clairvoyant": [], "lower bound": []}
    for s in range(22):
        solver.solve(torch.from_numpy(w.inputs).to(dev))
        torch.cuda.synchronize()
        cyc = np.abs(solver.get_profile()[:, 15]).astype(np.float64)
        if len(hist) >= 10 and s % 3 == 0:
            pred = np.maximum.reduce(hist[-10:])
            res["longest-first by max of last 10"].append(schedule(np.argsort(-pred, kind="stable"), cyc))
            res["multifit on max of last 10"].append(schedule(multifit_order(pred)[0], cyc))
            res["longest-first clairvoyant"].append(schedule(np.argsort(-cyc, kind="stable"), cyc))
            res["multifit clairvoyant"].append(schedule(multifit_order(cyc)[0], cyc))
            res["lower bound"].append(max(cyc.sum() / slots, cyc.max()))
        hist.append(cyc)
        w = perturb_workload(w, 7000 + 131 * s)
    for k, v in res.items():
        print(f"{k:34s}: mean {np.mean(v) / ghz / 1e6:.4f} ms over {len(v)} steps")
