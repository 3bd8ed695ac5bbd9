"""Off-line model of how the solve kernel's robots pack onto the wave slots, from per-robot section cycles
(tools/dump_schedule_data.py with the -DMPC_SECTION_PROFILE build).  Compares the one-job-per-robot launch (what the kernel
did up to round 2) with a persistent-wave kernel that treats ADMM and polish as separate jobs.

usage: python tools/sched_model.py gpurun_out/sched_data_prof.npz [GHz]"""
import heapq, sys
import numpy as np

d = np.load(sys.argv[1])
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.09
info, prof, kms = d["info"], d["prof"].astype(np.float64), d["kernel_ms"]
S, n, _ = prof.shape
slots = 1024
total = prof[:, :, 15]
polish = prof[:, :, 11:15].sum(2)           # set-up + H + refine + finish (the polish's K formation / sweep are inside 6 / 7)
nfact = info[:, :, 4].astype(np.float64)
# sections 6 (K formation) and 7 (sweep) cover all factorisations of the solve; the polish's share is one of them
per_fact = (prof[:, :, 6] + prof[:, :, 7]) / np.maximum(nfact, 1)
has_pol = info[:, :, 2] != 0
pol_job = np.where(has_pol, polish + per_fact, 0.0)
admm_job = total - pol_job


def lpt(order, cyc):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for r in order:
        t = heapq.heappop(free) + cyc[r]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


def split(order, a, p, reload, policy):
    """persistent waves: ADMM jobs in `order`; a robot's polish job (p + reload) becomes ready when its ADMM ends.
    policy 'admm_first': a free wave takes the next ADMM job while there is one, else the oldest ready polish job, else waits.
    policy 'own': the wave that finished the ADMM polishes at once while unstarted ADMM jobs remain above `keep`... (= the old kernel if always)"""
    ev = [(0.0, i) for i in range(slots)]    # (time free, wave)
    heapq.heapify(ev)
    q = list(order)[::-1]
    ready = []                               # (ready time, robot)
    end = 0.0
    pending = {}                             # wave -> robot whose ADMM it is running
    npol_left = int((p > 0).sum())
    waiting = []
    while ev:
        t, w = heapq.heappop(ev)
        if w in pending:
            r = pending.pop(w)
            if p[r] > 0:
                heapq.heappush(ready, (t, r))
        if q:
            r = q.pop()
            pending[w] = r
            heapq.heappush(ev, (t + a[r], w))
            end = max(end, t + a[r])
        elif ready:
            tr, r = heapq.heappop(ready)
            te = max(t, tr) + p[r] + reload
            npol_left -= 1
            heapq.heappush(ev, (te, w))
            end = max(end, te)
        elif pending:
            # nothing to do now: wake up when the next ADMM job ends
            tn = min(tt for tt, ww in ev if ww in pending) if any(ww in pending for _, ww in ev) else None
            if tn is not None:
                heapq.heappush(ev, (tn + 1e-9, w))
        # else: this wave retires
    return end


res = {}
for s in range(10, S):
    hist_tot = total[s - 10:s].max(0)
    hist_admm = admm_job[s - 10:s].max(0)
    c = total[s]
    res.setdefault("measured kernel ms", []).append(kms[s, 1])
    res.setdefault("lower bound", []).append(c.sum() / slots / ghz / 1e6)
    res.setdefault("one job/robot, LPT by max-10 (round 2)", []).append(lpt(np.argsort(-hist_tot, kind="stable"), c) / ghz / 1e6)
    res.setdefault("one job/robot, clairvoyant LPT", []).append(lpt(np.argsort(-c, kind="stable"), c) / ghz / 1e6)
    for reload in (6e3, 12e3):
        res.setdefault(f"split, ADMM first by max-10, reload {reload/1e3:.0f}k", []).append(split(np.argsort(-hist_admm, kind="stable"), admm_job[s], pol_job[s], reload, "admm_first") / ghz / 1e6)
    res.setdefault("split, clairvoyant ADMM order, reload 6k", []).append(split(np.argsort(-admm_job[s], kind="stable"), admm_job[s], pol_job[s], 6e3, "admm_first") / ghz / 1e6)
    res.setdefault("split, robot order, reload 6k", []).append(split(np.arange(n), admm_job[s], pol_job[s], 6e3, "admm_first") / ghz / 1e6)
for k, v in res.items():
    print(f"{k:52s}: mean {np.mean(v):.4f} ms  (min {np.min(v):.4f} max {np.max(v):.4f}, {len(v)} steps)")
print("mean cycles: total %.0f k, ADMM job %.0f k, polish job %.0f k; polish on %.3f of solves" % (total[10:].mean() / 1e3, admm_job[10:].mean() / 1e3, pol_job[10:][has_pol[10:]].mean() / 1e3, has_pol[10:].mean()))
