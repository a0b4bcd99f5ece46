"""Timeline of one k_cs_stream launch (MFM_CS_TRACE): per step, when each actor started / finished relative to the walker, and which
dependency released the walker last. usage: python scripts/cs_trace_analyze.py <trace file> [first step] [steps]"""
import sys

import numpy as np

f = sys.argv[1]
hdr = open(f).readline()
print(hdr.strip())
a = np.loadtxt(f, comments="#", dtype=np.int64)
na = a[:, 0].max() + 1
ns = a[:, 1].max() + 1
T = np.zeros((na, ns, 3))
T[a[:, 0], a[:, 1]] = a[:, 2:5] / 100.0  # us
NB = (na - 3) // 2
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else ns // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
Lw = int(hdr.split("Lw")[1].split()[0])
RD = int(hdr.split("RD")[1].split()[0])
base = T[0, s0, 0]
print("step | walker wait-start, start, end | X after-wait, end | Y begin, gated, end | U(min..max) after-wait, end | S(min..max) after-wait, end")
for s in range(s0, min(ns, s0 + n)):
    U, S = T[3:3 + NB, s], T[3 + NB:3 + 2 * NB, s]
    print("%4d | %7.2f %7.2f %7.2f | %7.2f %7.2f | %7.2f %7.2f %7.2f | %7.2f..%7.2f %7.2f..%7.2f | %7.2f..%7.2f %7.2f..%7.2f" % (
        s, *(T[0, s] - base), *(T[1, s, 1:] - base), *(T[2, s] - base), U[:, 1].min() - base, U[:, 1].max() - base, U[:, 2].min() - base,
        U[:, 2].max() - base, S[:, 1].min() - base, S[:, 1].max() - base, S[:, 2].min() - base, S[:, 2].max() - base))
# steady state: mean period, and mean lags along the loop walker(s - Lw) end -> X -> U -> S(s) -> Y(s) -> walker(s)
ss = np.arange(max(Lw + 2, ns // 4), ns - 2)
per = np.diff(T[0, ss, 1]).mean()
print("mean period %.2f us/step; walker busy %.2f" % (per, (T[0, ss, 2] - T[0, ss, 1]).mean()))
wl_end = T[0, ss - Lw, 2]
x_end = T[1, ss - Lw, 2]
u_end = T[3:3 + NB, :, 2][:, ss - Lw].max(axis=0)
u_aw = T[3:3 + NB, :, 1][:, ss - Lw].max(axis=0)
s_aw = T[3 + NB:, :, 1][:, ss].max(axis=0)
s_end = T[3 + NB:, :, 2][:, ss].max(axis=0)
y_g, y_end = T[2, ss, 1], T[2, ss, 2]
w_start = T[0, ss, 1]
print("loop (means, us): walker(s-Lw) end -> X end %.2f -> U wake (max over ranges) %.2f -> U end (max) %.2f -> S wake (max) %.2f -> S end (max) %.2f -> Y gated %.2f "
      "-> Y end %.2f -> walker(s) start %.2f" % ((x_end - wl_end).mean(), (u_aw - x_end).mean(), (u_end - u_aw).mean(), (s_aw - u_end).mean(),
                                                   (s_end - s_aw).mean(), (y_g - s_end).mean(), (y_end - y_g).mean(), (w_start - y_end).mean()))
print("slack of S(s): S end (max) - walker(s) start: %.2f; Y(s) begin - walker(s-RD) end: %.2f" % ((s_end - w_start).mean(), (T[2, ss, 0] - T[0, ss - RD, 2]).mean()))
