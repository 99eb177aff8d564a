"""Prints the kernels of the last default-mode frame of a tools/kernel_times.sh
trace (gpurun_out/kt_<tag>/*.db) in launch order with their durations."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/*.db")[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_render<3" in r[0] or "k_render<1" in r[0]]
i0 = idx[-1]
t0 = rows[i0][1]
total = 0
for n, s, e in rows[i0:]:
    if "rt::" not in n:
        break
    print("%9.3f ms  +%7.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n[:52]))
    total = (e - t0) / 1e6
    if "k_compose" in n:
        break
print("frame: %.3f ms first launch to last end" % total)
