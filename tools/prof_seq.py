"""Dump the kernel sequence of the last training step of a rocprofv3 kernel trace (developer tool)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
sel = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(sel[0]['Start_Timestamp']); prev = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::|at::native::|void ', '', r['Kernel_Name'])[:90]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
    prev = e
