import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None;acc=[]
for r in rows+[None]:
    name=r['Kernel_Name'][:40] if r else None
    if name!=prev and prev is not None:
        if 'at::native' not in prev and 'split_b' not in prev:
            acc.sort(); print("%-42s n=%2d median %.1f us min %.1f"%(prev,len(acc),acc[len(acc)//2],acc[0]))
        acc=[]
    prev=name
    if r: acc.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
