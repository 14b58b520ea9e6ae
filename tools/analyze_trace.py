"""Kernel-trace timeline summary: python tools/analyze_trace.py <rocprofv3 -d dir> [from-fraction to-fraction]: union-busy time, per-kernel totals, first 60 launches of the window."""
import csv,sys,glob,collections
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0].split('<')[0][:40],r.get('Queue_Id'),r.get('Stream_Id')) for r in rows]
ev.sort()
# take the last 40% of time as steady state
t0=ev[0][0]; t1=max(e[1] for e in ev)
lo=t0+(t1-t0)*float(sys.argv[2]) if len(sys.argv)>2 else t0+(t1-t0)*0.6
hi=t0+(t1-t0)*float(sys.argv[3]) if len(sys.argv)>3 else t1
sel=[e for e in ev if e[0]>=lo and e[1]<=hi]
span=max(e[1] for e in sel)-min(e[0] for e in sel)
busy=0; cur_s=None; cur_e=None
for s,e,_,_,_ in sel:
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
tot=sum(e[1]-e[0] for e in sel)
print('span us',span/1e3,'union busy us',busy/1e3,'sum kernel us',tot/1e3,'queues',collections.Counter(e[3] for e in sel))
by=collections.defaultdict(lambda:[0,0])
for s,e,n,_,_ in sel:
    by[n][0]+=1; by[n][1]+=e-s
for n,(c,t) in sorted(by.items(),key=lambda x:-x[1][1])[:12]:
    print('%-42s %5d %9.1f us avg %7.1f'%(n,c,t/1e3,t/1e3/c))
# print a short timeline
for s,e,n,q,st in sel[:60]:
    print('%9.1f %7.1f %s q%s'%((s-sel[0][0])/1e3,(e-s)/1e3,n,q))
