import csv,glob,sys
f=glob.glob(sys.argv[1]+'/runc/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'chain' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
n=len(d)//4
for i,name in enumerate(['pow','divide','add','as_type']):
    seg=d[i*n:(i+1)*n]
    print(sys.argv[1].split('/')[-1], name, len(seg), 'avg %.1f min %.1f max %.1f'%(sum(seg)/len(seg), min(seg), max(seg)), rows[i*n]['Kernel_Name'][:40])
