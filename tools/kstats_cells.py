import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
    if not f: print(d,'no stats'); continue
    print('==',d)
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'cell' in n: print('  %-40s calls %4s avg %8.1f us min %8.1f' % (n.split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
