"""Probe: can two controller ticks (one with, one without the MPC update) be captured in a HIP graph and replayed?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
from rl_mpc_locomotion_amd.synthetic import TickStream

n = 4096
ts = TickStream(n, seed=1, config=2)
def run(use_graph, ticks=200):
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10)
    ins = [tuple(torch.from_numpy(a).cuda() for a in ts.tick(k)) for k in range(ticks)]
    sd, sb, sc = (torch.empty_like(x) for x in ins[0])
    sd2, sb2, sc2 = (torch.empty_like(x) for x in ins[0])
    out = torch.zeros((ticks, n, 12), dtype=torch.float32, device="cuda")
    for k in range(4):                       # warm-up (also: first tick is a cold solve)
        ctl.run(*ins[k])
    torch.cuda.synchronize()
    g = None
    if use_graph:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                t0_ = ctl.run(sd, sb, sc).clone()
                t1_ = ctl.run(sd2, sb2, sc2).clone()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4, ticks, 2):
        if use_graph:
            sd.copy_(ins[k][0]); sb.copy_(ins[k][1]); sc.copy_(ins[k][2])
            sd2.copy_(ins[k + 1][0]); sb2.copy_(ins[k + 1][1]); sc2.copy_(ins[k + 1][2])
            g.replay()
            out[k].copy_(t0_); out[k + 1].copy_(t1_)
        else:
            out[k].copy_(ctl.run(*ins[k])); out[k + 1].copy_(ctl.run(*ins[k + 1]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / (ticks - 4) * 1e3, out
ms_e, oe = run(False)
try:
    ms_g, og = run(True)
    print("eager %.4f ms/tick, graph %.4f ms/tick, identical torques: %s" % (ms_e, ms_g, bool(torch.equal(oe[8:], og[8:]))))
except Exception as e:
    print("eager %.4f ms/tick; graph capture failed: %r" % (ms_e, e))
