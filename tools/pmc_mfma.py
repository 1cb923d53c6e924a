"""Fold a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_BUSY_CU_CYCLES; --kernel-trace only) into
profiles/<name>.json: per kernel, matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel
cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back).  Developer tool.
usage: pmc_mfma.py <pmc_dir> <out.json> kernel_substring [...]"""
import collections, csv, glob, json, sys


def main():
    d, out = sys.argv[1:3]
    want = sys.argv[3:]
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
        n[r['Kernel_Name']].add(r['Dispatch_Id'])
    res = {}
    for w in want:
        ks = [k for k in tot if w in k]
        if not ks:
            continue
        k = ks[0]
        c = tot[k]
        launches = max(1, len(n[k]))
        cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / launches
        mfma = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / launches
        res[w] = {"launches": launches, "kernel_cycles": round(cyc), "SQ_VALU_MFMA_BUSY_CYCLES": round(mfma),
                  "SQ_BUSY_CU_CYCLES": round(c.get('SQ_BUSY_CU_CYCLES', 0.0) / launches),
                  "mfma_busy_share_of_1024_simds": round(mfma / (1024.0 * cyc), 4) if cyc else None}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python3 "
                         "bench.py --steps 3 --warmup 2 --no-cpu-baseline",
               "note": "mfma_busy_share = matrix-pipe busy cycles summed over the chip / (1024 SIMDs x kernel cycles); a dense "
                       "MFMA stream reaches ~1.0",
               "kernels": res}, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
