"""Timeline of one TV-L1 flow from a rocprofv3 --kernel-trace CSV: busy time, gaps, launches per kernel.
   python tools/tv_timeline.py <..._kernel_trace.csv>   (trace of tools/tvl1_time.py: the LAST flow of the run is taken)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_tv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into flows at k_tv_init_minmax
starts = [i for i, r in enumerate(rows) if "k_tv_init_minmax" in r["Kernel_Name"]]
last = rows[starts[-1]:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(last, last[1:])]
print(f"launches {len(last)}  span {(t1 - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  gaps {sum(gaps) / 1e3:.1f} us "
      f"(>5us: {sum(1 for g in gaps if g > 5000)} gaps, {sum(g for g in gaps if g > 5000) / 1e3:.1f} us)")
per = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k][0] += 1
    per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:44s} n={n:4d} total {t / 1e3:8.1f} us  mean {t / n / 1e3:7.1f} us")
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us):", [round(g / 1e3, 1) for g in big])
