"""Per-tick durations of the run's LAST day from a rocprofv3 kernel trace (VDS_RUN_GROUPS=1: one launch per tick and kernel):
walk / tick in us, '*' where k_tick_dense ran its 16-lane form (256-entry tables) in that slot.   python profiles/pertick.py <dir>"""
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def series(pat):
    k=[r for r in rows if pat in r['Kernel_Name']]
    k.sort(key=lambda r:int(r['Start_Timestamp']))
    k=k[-148:]
    return [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in k], ['*' if ', 256, ' in r['Kernel_Name'] else ' ' for r in k]
W,_=series('k_dfs_walk'); T,F=series('k_tick_dense')
if not T: T,F=series('k_tick_rows')
if not W: W=[0.0]*len(T)
print("walk day ms %.2f tick day ms %.2f (mean per launch %.2f us; %d of %d slots in the 16-lane form)"%(sum(W)/1000,sum(T)/1000,sum(T)/max(1,len(T)),F.count('*'),len(T)))
for t in range(0,len(T),8): print("%3d"%t, ' '.join("%4.0f/%3.0f%s"%(W[t+i],T[t+i],F[t+i]) for i in range(min(8,len(T)-t))))
