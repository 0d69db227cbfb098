"""Compact timeline of the last bench step from a rocprofv3 kernel trace (diagnostic).
usage: python tools/timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('p2::', ''), r['Stream_Id']) for r in rows]
ev.sort()
# the stage-0 sampler (NBL = 2) runs twice per step: stratified, then resumed for the transition
big = [i for i, e in enumerate(ev) if e[2].startswith('fps_bucket_kernel<2')]
first_of_step = big[-2]
# the step begins with the set-up kernels a little before it: walk back over events closer than 200 us
b = first_of_step
while b > 0 and ev[b][0] - ev[b - 1][1] < 200000: b -= 1
step = ev[b:]
t0 = step[0][0]
print('step events', len(step), 'span ms', (max(e[1] for e in step) - t0) / 1e6)
for sid in sorted(set(e[3] for e in step)):
    print('--- stream', sid)
    cur = None
    for s, e, n, st in step:
        if st != sid: continue
        n = n[:40]
        if cur and cur[2] == n and s - cur[1] < 30000:
            cur[1] = e; cur[3] += 1; cur[4] += e - s
        else:
            if cur: print('  %8.2f -> %8.2f ms  x%-3d busy %7.2f ms  %s' % ((cur[0] - t0) / 1e6, (cur[1] - t0) / 1e6, cur[3], cur[4] / 1e6, cur[2]))
            cur = [s, e, n, 1, e - s]
    if cur: print('  %8.2f -> %8.2f ms  x%-3d busy %7.2f ms  %s' % ((cur[0] - t0) / 1e6, (cur[1] - t0) / 1e6, cur[3], cur[4] / 1e6, cur[2]))
