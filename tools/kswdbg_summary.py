import re, collections, sys
d = collections.defaultdict(lambda: [0, 0, 0.0, 0.0, 0.0])
for l in open(sys.argv[1]):
    m = re.match(r"KSW class (\d) tasks (\d+) cells (\S+) max (\S+) \(q (\d+) t (\d+)\) ms (\S+)", l)
    if m:
        k = int(m.group(1)); e = d[k]; e[0] += 1; e[1] += int(m.group(2)); e[2] += float(m.group(3)); e[3] = max(e[3], float(m.group(4))); e[4] += float(m.group(7))
for k, e in sorted(d.items()):
    print("class", k, "launches", e[0], "tasks", e[1], "cells %.3g" % e[2], "maxcells %.3g" % e[3], "ms %.0f" % e[4], "GCUPS %.1f" % (e[2] / e[4] / 1e6))
