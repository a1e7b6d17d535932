import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
key = sys.argv[2] if len(sys.argv) > 2 else 'poisson3d_q1_cf'
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
tot = 0
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot += d
    n = r['Kernel_Name'].split('(')[0][-70:]
    agg[n][0] += d; agg[n][1] += 1
    if d > int(sys.argv[3]) if len(sys.argv) > 3 else 15000:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  {d/1e3:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>7}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}  {n}")
print('kernels', len(step), 'sum', tot / 1e3, 'span', (int(step[-1]['End_Timestamp']) - t0) / 1e3)
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"   {d/1e3:9.1f} us {c:3d} calls  {n}")
