import csv, collections, sys
for d in sys.argv[1:]:
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if 'igemm' in r['Kernel_Name']:
            agg[r['Kernel_Name'][28:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k)
        for c,vals in sorted(v.items()): print('   %-34s n=%d mean=%.5g'%(c,len(vals),sum(vals)/len(vals)))
