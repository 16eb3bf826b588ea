import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('k_mm_','k_glue','k_reward'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n0=int(sys.argv[2]) if len(sys.argv)>2 else 600
base=int(rows[n0]['Start_Timestamp'])
for r in rows[n0:n0+14]:
    st=int(r['Start_Timestamp'])-base; en=int(r['End_Timestamp'])-base
    print("%-14s q%s start %8.2f end %8.2f dur %6.2f" % (r['Kernel_Name'].split('(')[0].replace('void pilco::','').replace('pilco::','')[:14], r.get('Queue_Id','?'), st/1e3,en/1e3,(en-st)/1e3))
