import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/t/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"][:14] if "Direction" in r else r.get("Name", "copy")[:14]))
for f in glob.glob(root + "/t/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"].split("(")[0][-22:]))
ev.sort()
if not ev:
    print("no events"); sys.exit()
# last burst: events after the last gap > 20 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 20_000_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
print("last call: %d events, span %.2f ms" % (len(ev), (max(e[1] for e in ev) - t0) / 1e6))
kinds = {}
for s, e, k in ev:
    g = "H2D" if "HOST_TO" in k.upper() or "H2D" in k.upper() else "D2H" if "TO_HOST" in k.upper() or "D2H" in k.upper() else "kernel" if k.startswith("K:") else k
    kinds.setdefault(g, []).append((s - t0, e - t0))
for g, iv in kinds.items():
    busy = sum(e - s for s, e in iv)
    print("%-8s n=%4d busy %.2f ms, first start %.2f, last end %.2f" % (g, len(iv), busy / 1e6, iv[0][0] / 1e6, max(e for s, e in iv) / 1e6))
# coarse timeline: per 0.5 ms bucket which kinds are active
import math
span = max(e[1] for e in ev) - t0
nb = int(math.ceil(span / 500_000))
for b in range(nb):
    lo, hi = b * 500_000, (b + 1) * 500_000
    row = []
    for g, iv in kinds.items():
        act = sum(max(0, min(e, hi) - max(s, lo)) for s, e in iv) / 500_000.0
        row.append("%s %.1f" % (g, act))
    print("%5.1f ms: %s" % (lo / 1e6, "  ".join(row)))
