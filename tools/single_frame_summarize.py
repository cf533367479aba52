import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
gi=[i for i,n in enumerate(names) if n.startswith('k_gen_primary')][-3]
j=gi; prev=int(rows[gi]['Start_Timestamp']); tk=tg=0
while True:
    r=rows[j]; s_=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:44]:44s} dur {(e-s_)/1e3:8.1f} us gap {(s_-prev)/1e3:6.1f} us grid {r.get('Grid_Size','?')}")
    tk+=e-s_; tg+=max(s_-prev,0); prev=e
    if r['Kernel_Name'].startswith('k_final_draw'): break
    j+=1
print('kernels %.3f ms, gaps %.3f ms'%(tk/1e6,tg/1e6))
