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
# steady state (the helpers' stamps exist for EVEN steps only: wavefront pairs take the steps in turn, the first of each pair stamps)
ss = np.arange(max(Lw + 2, ns // 4), ns - 2)
per = np.diff(T[0, ss, 1]).mean()
print("mean period %.2f us/step; walker busy %.2f, waits %.2f" % (per, (T[0, ss, 2] - T[0, ss, 1]).mean(), (T[0, ss, 1] - T[0, ss, 0]).mean()))
ev = ss[(ss % 2 == 0) & ((ss - Lw) % 2 == 0)] if Lw % 2 == 0 else None
def lag(name, a, b):
    print("  %-46s %6.2f" % (name, (b - a).mean()))
se = ss[ss % 2 == 0]
print("chain of an even step s (means, us):")
lag("walker(s) end -> X(s) sees it", T[0, se, 2], T[1, se, 1])
lag("X(s): reads, stores, drain, publish", T[1, se, 1], T[1, se, 2])
lag("X(s) end -> U(s) wakes (max over ranges)", T[1, se, 2], T[3:3 + NB, :, 1][:, se].max(axis=0))
lag("U(s) wake -> end (max over ranges)", T[3:3 + NB, :, 1][:, se].max(axis=0), T[3:3 + NB, :, 2][:, se].max(axis=0))
if Lw % 2 == 0:
    sl = se[se + Lw < ns]
    lag("U(s) end (max) -> S(s + Lw) wakes (max)", T[3:3 + NB, :, 2][:, sl].max(axis=0), T[3 + NB:, :, 1][:, sl + Lw].max(axis=0))
lag("S(s) wake -> end (max over ranges)", T[3 + NB:, :, 1][:, se].max(axis=0), T[3 + NB:, :, 2][:, se].max(axis=0))
lag("S(s) end (max) -> Y(s) has its data", T[3 + NB:, :, 2][:, se].max(axis=0), T[2, se, 1])
lag("Y(s) data -> staged", T[2, se, 1], T[2, se, 2])
lag("Y(s) staged -> walker(s) starts", T[2, se, 2], T[0, se, 1])
lag("walker(s - 1) end -> walker(s) starts", T[0, se - 1, 2], T[0, se, 1])
lag("S(s) begin -> wake (waits for U)", T[3 + NB:, :, 0][:, se].max(axis=0), T[3 + NB:, :, 1][:, se].max(axis=0))
lag("U(s) begin -> wake (waits for X)", T[3:3 + NB, :, 0][:, se].max(axis=0), T[3:3 + NB, :, 1][:, se].max(axis=0))
lag("Y(s) begin -> data (waits)", T[2, se, 0], T[2, se, 1])
