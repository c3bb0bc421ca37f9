"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) per kernel/grid."""
import csv, collections, sys, glob, os
def main(dirs):
    for d in dirs:
        for f in glob.glob(os.path.join(d, '*_counter_collection.csv')):
            rows = list(csv.DictReader(open(f)))
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            meta = {}
            for r in rows:
                k = (r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:64], r['Grid_Size'])
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                meta[k] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
            for k, v in agg.items():
                print(f, k, 'vgpr/agpr/lds/scratch', meta[k])
                for c, x in sorted(v.items()):
                    print(f"    {c:32s} {sum(x)/len(x):16.1f}")
        for f in glob.glob(os.path.join(d, '*_kernel_trace.csv')):
            rows = list(csv.DictReader(open(f)))
            dd = collections.defaultdict(list)
            for r in rows:
                dd[(r['Kernel_Name'][:70], r.get('Grid_Size_X') or r.get('Grid_Size'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            for k, v in dd.items():
                print(f, k, 'calls', len(v), 'avg_us %.2f' % (sum(v) / len(v) / 1e3))
if __name__ == '__main__':
    main(sys.argv[1:])
