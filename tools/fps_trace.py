"""Reads the per-wave phase times dumped by P2_FPS_STAMPS=1 P2_FPS_TRACE=<file> (fps_bucket.hip) and reports, per
step, which wave reached the step barrier last and how its time divides (diagnostic)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 8).astype(np.int64)
t = t[(t[:, :, 0] > 0).all(axis=1)]
print('steps', len(t))
start = t[:, :, 0].min(axis=1)                 # first wave to start the step
test = t[:, :, 1] - t[:, :, 0]
upd = t[:, :, 2] - t[:, :, 1]
red = t[:, :, 3] - t[:, :, 2]
arrive = t[:, :, 3]
last = arrive.argmax(axis=1)
k = t[:, :, 7]
idx = np.arange(len(t))
step_len = np.diff(t[:, 0, 0])
print('step length: mean %.0f  median %.0f  p90 %.0f cycles' % (step_len.mean(), np.median(step_len), np.percentile(step_len, 90)))
print('last wave to reach the barrier: its test %.0f, update %.0f, reduce %.0f cycles; touched buckets %.2f (all waves: %.2f)' %
      (test[idx, last].mean(), upd[idx, last].mean(), red[idx, last].mean(), k[idx, last].mean(), k.mean()))
print('start skew of the last wave vs the first starter: %.0f' % (t[idx, last, 0] - start).mean())
print('barrier release after the last arrival: %.0f' % (t[:, :, 4].min(axis=1) - arrive.max(axis=1)).mean())
print('final phase (release -> second barrier arrival of wave 0): %.0f' % (t[:, 0, 5] - t[:, 0, 4]).mean())
print('second barrier + sample read (wave 0 arrival -> all waves past it): %.0f' % (t[:, :, 6].max(axis=1) - t[:, 0, 5]).mean())
for kk in range(0, 5):
    sel = k[idx, last] == kk
    if sel.sum():
        print('  last wave had %d touched: %5.1f%% of steps, its test+update+reduce %.0f' % (kk, 100 * sel.mean(), (arrive[idx, last] - t[idx, last, 0])[sel].mean()))
print('per-k update time over all waves:', {int(kk): int(upd[k == kk].mean()) for kk in range(0, 5) if (k == kk).any()})
