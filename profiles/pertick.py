import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def series(pat):
    k=[r for r in rows if pat in r['Kernel_Name']]
    k.sort(key=lambda r:int(r['Start_Timestamp']))
    return [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in k][-148:]
W=series('k_dfs_walk') or [0.0]*148; T=series('k_tick_dense') or series('k_tick_rows')
print("walk day ms %.2f tick day ms %.2f"%(sum(W)/1000,sum(T)/1000))
for t in range(0,148,8): print("%3d"%t, ' '.join("%4.0f/%3.0f"%(W[t+i],T[t+i]) for i in range(min(8,148-t))))
