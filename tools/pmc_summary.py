"""Summarise rocprofv3 --pmc passes (one sub-directory per pass) of tools/pmc_conv.py as a markdown table.
usage: python tools/pmc_summary.py <dir>"""
import csv, collections, glob, sys
d = sys.argv[1]
tab = collections.OrderedDict()
for f in sorted(glob.glob(d + '/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'igemm' not in r['Kernel_Name']: continue
        name = r['Kernel_Name'].split('igemm_')[1].split('(')[0]
        key = (int(r['Dispatch_Id']), name, r['Grid_Size'])
        tab.setdefault(key, collections.OrderedDict())
        tab[key][r['Counter_Name']] = tab[key].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
for key, c in tab.items():
    print('dispatch %d  igemm_%s  grid %s' % key)
    for k, v in c.items(): print('    %-28s %.4g' % (k, v))
