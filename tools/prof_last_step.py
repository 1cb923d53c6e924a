"""Summarise the last training step of a rocprofv3 kernel trace (developer tool)."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
sel = rows[idx[-2] + 1: idx[-1] + 1]
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
def cat(n):
    if n.startswith('Cijk'): return 'gemm_f32(lib)' if '_SB_' in n else 'gemm_lp(lib)'
    for k in ('joint_', 'rnnt_', 'gemm_bf16', 'layernorm_kernel', 'glu_dwconv', 'bn_silu', 'bn_running', 'cl_', 'adamw', 'relpos'):
        if k in n: return 'HIP:' + k
    if 'LSTM' in n: return 'lstm'
    if 'conv' in n.lower() or 'Im2d' in n or 'Col2Im' in n or 'igemm' in n or 'transpose' in n.lower(): return 'conv(miopen)'
    if 'ctc' in n: return 'ctc'
    if 'layer_norm' in n: return 'layernorm(aten)'
    if 'softmax' in n.lower(): return 'softmax'
    if 'copy' in n.lower(): return 'copy/cast'
    if 'dropout' in n or 'masked_scale' in n: return 'dropout'
    if 'reduce_kernel' in n: return 'reduce'
    if 'elementwise' in n: return 'elementwise'
    return 'other'
agg = collections.defaultdict(lambda: [0, 0.0]); names = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    c = cat(r['Kernel_Name']); agg[c][0] += 1; agg[c][1] += d
    names[r['Kernel_Name']][0] += 1; names[r['Kernel_Name']][1] += d
tot = sum(v[1] for v in agg.values())
print(f"last step: {len(sel)} kernels, span {span:.2f} ms, busy {tot/1e3:.2f} ms")
for c, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d/1e3:8.3f} ms x{n:5d}  {c}")
print("--- top kernels")
for n, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"{d/1e3:8.3f} ms x{c:4d} avg {d/c:8.1f} us  {n[:100]}")
