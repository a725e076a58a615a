import csv, sys, collections
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r['Kernel_Name']
            if 'k_fuse<' not in k: continue
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(path.split('/')[-2], k)
        for c, v in d.items():
            print('   %-28s mean/launch %.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
