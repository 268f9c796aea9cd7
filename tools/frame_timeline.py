import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(s, e, n.replace("void ", "").replace("srrg2amd::", "").replace("(anonymous namespace)::", "").split("(")[0]) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
# frames: delimited by k_clip_flag
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_clip_flag")]
fr = rows[idx[-3]:idx[-2]]
t0 = fr[0][0]
busy = 0
prev_end = t0
for s, e, n in fr:
    print("%8.1f +%6.1f gap %7.1f us  %s" % ((s - t0) / 1000, (s - prev_end) / 1000, (e - s) / 1000, n[:60]))
    busy += e - s
    prev_end = e
print("frame span %.1f us, kernels %.1f us" % ((idx and (rows[idx[-2]][0] - t0) / 1000), busy / 1000))
