"""Summarise rocprofv3 --pmc CSVs (tools/pmc_sq.sh, tools/profile_round.sh): per kernel INSTANCE and
launch shape (full template name, grid size, LDS bytes), the mean of every counter over its dispatches.
A temporal bench run launches the same instance in two shapes (the one spatial call that builds the
previous frame has another halo / grid): they are separate entries, so that the steady-state launches
are not averaged with the warm-up one.
   python tools/pmc_summary.py gpurun_out/<tag>"""
import csv, glob, os, sys, collections


def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "")


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "p*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            key = f"{short(r['Kernel_Name'])} grid={r['Grid_Size']} lds={r['LDS_Block_Size']}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")


if __name__ == "__main__":
    main(sys.argv[1])
