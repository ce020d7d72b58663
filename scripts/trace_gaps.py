import csv, glob
rows=[]
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ours=[r for r in rows if any(k in r["Kernel_Name"] for k in ("e1_kernel","k01_kernel","pairs_kernel","sum_pairs_kernel"))]
ours.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 400 kernels (steady state)
ours=ours[-400:]
import collections
gaps=collections.defaultdict(list); durs=collections.defaultdict(list)
def short(n):
    for k in ("e1_kernel","k01_kernel","pairs_kernel","sum_pairs_kernel"):
        if k in n: return k
for a,b in zip(ours,ours[1:]):
    gaps[short(a["Kernel_Name"])+"->"+short(b["Kernel_Name"])].append((int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3)
for r in ours: durs[short(r["Kernel_Name"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
import statistics as st
for k,v in durs.items(): print("dur %-18s median %.1f us (n=%d)"%(k,st.median(v),len(v)))
for k,v in gaps.items(): print("gap %-36s median %.1f us (n=%d)"%(k,st.median(v),len(v)))
