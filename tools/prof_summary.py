"""Summarise rocprofv3 CSV output of tools/prof.sh: per-kernel mean duration and counters."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def short(n):
    n = n.split('(')[0]
    import re as _re
    return _re.sub(r'(void )?pbh(32|64)?::', '', n)
# kernel trace
for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('== kernel trace (us): name, calls, mean, min')
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f'{k:40s} {len(v):4d} {sum(v)/len(v):10.1f} {min(v):10.1f}')
for sub in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f'== {sub}: mean counter value per dispatch')
        for k, cs in d.items():
            if not (k.startswith('k_')):
                continue
            print(f'{k:40s} ' + ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(cs.items())))
