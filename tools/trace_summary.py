import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        n = r['Kernel_Name']
        for k in ('msda_bwd_prepare','plan_cells','msda_bwd_cell_sort','msda_fwd_vec','msda_bwd_tile_reduce','msda_bwd_vec','msda_taps_coarse'):
            if k in n:
                by[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
    print(d)
    tot = 0
    for k, v in by.items():
        v.sort(); dd = [x[1] for x in v][-100:]
        tot += sum(dd)/len(dd)/1e3
        print('  %-24s n=%d last100 mean %.1f us' % (k, len(v), sum(dd)/len(dd)/1e3))
    print('  sum %.1f us' % tot)
