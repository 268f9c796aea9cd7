import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(s, e, n.replace("void ", "").replace("srrg2amd::", "").replace("(anonymous namespace)::", "").split("(")[0]) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
ends = [i for i, r in enumerate(rows) if "final" in r[2]]
# the last compute() that contains a grid kernel k_icp_step< or k_icp_step_fused
for a, b in reversed(list(zip(ends[:-1], ends[1:]))):
    sq = rows[a + 1:b + 1]
    if any(r[2].startswith(("k_icp_step<", "k_icp_step_fused")) for r in sq):
        t0 = sq[0][0]; prev = t0
        for s, e, n in sq:
            print("%8.1f +%5.1f %7.1f us  %s" % ((s - t0) / 1000, (s - prev) / 1000, (e - s) / 1000, n[:60])); prev = e
        print("span %.1f us" % ((sq[-1][1] - t0) / 1000))
        break
