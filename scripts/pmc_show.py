import csv, collections, sys, glob
for d in sorted(glob.glob(sys.argv[1])):
    rows = list(csv.DictReader(open(d + '/p_counter_collection.csv')))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if 'conv_igemm' in r['Kernel_Name'] or (len(sys.argv) > 2 and sys.argv[2] in r['Kernel_Name']):
            agg[(r['Kernel_Name'][:70], r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, k)
        for c, vals in sorted(v.items()):
            print('   %-32s %16.0f (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
