import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'hmmsort' in r['Name']:
        print(r['Name'][:44].ljust(46), r['Calls'], '%.3f ms'%(float(r['AverageNs'])/1e6), r['Percentage'])
