"""compare two outputs of tools/detect_probe.py"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for i in range(len(a['ncp'])):
    o, n1, n2 = int(a['ev_off'][i]), int(a['ncp'][i]), int(b['ncp'][i])
    x, y = a['cpts'][o:o + n1], b['cpts'][o:o + n2]
    if n1 != n2 or not np.array_equal(x, y):
        bad += 1
        if bad <= 5:
            k = int(np.flatnonzero(x[:min(n1, n2)] != y[:min(n1, n2)])[0]) if min(n1, n2) and (x[:min(n1, n2)] != y[:min(n1, n2)]).any() else -1
            print('read', i, 'fused flag', int(a['fused'][i]), 'n', n1, n2, 'first diff at', k, x[max(k - 3, 0):k + 4], y[max(k - 3, 0):k + 4],
                  'sorted?', bool(np.all(np.diff(x) > 0)), 'n diffs', int((x[:min(n1, n2)] != y[:min(n1, n2)]).sum()))
print('reads that differ:', bad, 'of', len(a['ncp']), '| fused', int(a['fused'].sum()), int(b['fused'].sum()))
