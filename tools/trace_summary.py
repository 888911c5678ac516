import csv,collections,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
seq=[]
for r in rows:
    n=r['Kernel_Name']; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    short=n.split('(')[0].replace('void ','').replace('aoc64::','').replace('aoc32::','f32::')
    if short.startswith('k_'): seq.append((short,d,int(r['Start_Timestamp']),int(r['End_Timestamp']), r.get('VGPR_Count'), r.get('Grid_Size_X')))
it=-1; out=collections.defaultdict(list)
for s in seq:
    if s[0].startswith('k_backward'): it+=1
    if it>=0: out[it].append(s)
sel=[int(a) for a in sys.argv[2:]] or [2,6,11]
for i in sel:
    print('--- iteration',i-2)
    t0=out[i][0][2]
    agg=collections.OrderedDict()
    for s in out[i]:
        if s[1]>20 or not s[0].startswith('k_ls'):
            print(f"  {s[0]:28s} {s[1]:9.1f} us  start+{(s[2]-t0)/1e3:9.1f}  vgpr={s[4]} grid={s[5]}")
        agg[s[0]]=agg.get(s[0],0)+s[1]
    print('   sum by kernel:',{k:round(v,1) for k,v in agg.items()}, 'span', (out[i][-1][3]-t0)/1e3)
