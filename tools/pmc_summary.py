"""Summarise rocprofv3 --pmc CSVs written by tools/pmc_run.sh: per kernel, the
mean of every counter over the dispatches of the timed steps.
   python tools/pmc_summary.py gpurun_out/<tag>"""
import csv, glob, os, sys, collections

def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "")

def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "p*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")

if __name__ == "__main__":
    main(sys.argv[1])
