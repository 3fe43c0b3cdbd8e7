#!/usr/bin/env python3
"""Timeline of the last host-pipeline call in a rocprofv3 kernel + memory-copy trace (tools/e2eprof.sh, e2eprof_decode.sh):
per kind of device event (SDMA copies, blit-kernel copies, coder kernels, gather / scatter) the periods in which it was
busy, in ms from the call's first device event.   tools/e2e_trace_summary.py <dir with the *_trace.csv files>"""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + '/*memory_copy_trace.csv'))[-1]
g = sorted(glob.glob(d + '/*kernel_trace.csv'))[-1]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'SDMA copy (%s)' % r['Direction'][12:].lower()))
for k in csv.DictReader(open(g)):
    n = k['Kernel_Name']
    nm = ('encode kernel' if 'dega_encode' in n else 'decode kernel' if 'dega_decode' in n else 'blit-kernel copy' if 'copyBuffer' in n else
          'gather' if 'gather' in n else 'scatter' if 'scatter' in n else None)
    if nm:
        ev.append((int(k['Start_Timestamp']), int(k['End_Timestamp']), nm))
ev.sort()
t0, hi = ev[0][0], ev[0][1]
for s_, e_, _ in ev:  # the last call begins after the last idle gap of 15 ms or more
    if s_ - hi > 15e6:
        t0 = s_
    hi = max(hi, e_)
c = [e for e in ev if e[0] >= t0]
print('last call of the trace: %.1f ms from its first to its last device event' % ((max(e[1] for e in c) - t0) / 1e6))
kinds = []
for e in c:
    if e[2] not in kinds:
        kinds.append(e[2])
for kind in kinds:
    iv = sorted((e[0], e[1]) for e in c if e[2] == kind and e[1] - e[0] > 50e3)  # (without the copies of a few bytes)
    if not iv:
        continue
    m = []
    for s_, e_ in iv:
        if m and s_ <= m[-1][1] + 0.25e6:
            m[-1][1] = max(m[-1][1], e_)
        else:
            m.append([s_, e_])
    print('  %-32s %3d events, busy (ms): %s' % (kind, len(iv), '  '.join('%.1f-%.1f' % ((a - t0) / 1e6, (b - t0) / 1e6) for a, b in m)))
