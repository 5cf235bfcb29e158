"""Prints the kernel timeline (start, end in us, queue, name) around the last ExSUM steps of a rocprofv3 --kernel-trace CSV:
python tools/parse_trace.py "gpurun_out/prof_dist/*/*_kernel_trace.csv" -- used to see what fills the gap between two streaming
kernels in the N > 1 (all-reduce) flow of bench.py."""
import csv, sys, glob
f=sorted(glob.glob(sys.argv[1]))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find last 40 kernels around exsum steps
idx=[i for i,r in enumerate(rows) if "k_exsum<" in r["Kernel_Name"]]
sel=idx[-6:]
t0=int(rows[sel[0]]["Start_Timestamp"])
for i in range(sel[0], sel[-1]+1):
    r=rows[i]
    n=r["Kernel_Name"].split("(")[0][-50:]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  q={r.get("Queue_Id","?")} {n}')
