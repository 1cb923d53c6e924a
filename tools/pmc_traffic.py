"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) into profiles/<name>.json: HBM bytes per launch
per kernel, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md (developer tool).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> kernel_substring [...]"""
import collections, csv, glob, json, sys


def per_launch(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        tot[r['Kernel_Name']] += float(r['Counter_Value'])
        n[r['Kernel_Name']].add(r['Dispatch_Id'])
    return {k: tot[k] / max(1, len(n[k])) for k in tot}


def main():
    fd, wd, out = sys.argv[1:4]
    want = sys.argv[4:]
    fe, wr = per_launch(fd, 'FETCH_SIZE'), per_launch(wd, 'WRITE_SIZE')
    res = {}
    for w in want:
        fk = [k for k in fe if w in k]
        if not fk:
            continue
        k = fk[0]
        res[w] = {"FETCH_SIZE_KiB": round(fe[k], 2), "WRITE_SIZE_KiB": round(wr.get(k, 0.0), 2),
                  "hbm_bytes_per_launch": int((2 * fe[k] + wr.get(k, 0.0)) * 1024)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 "
                         "bench.py --steps 2 --warmup 2 --no-cpu-baseline",
               "units": "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE "
                        "reports half of a wide coalesced read stream -> doubled; WRITE_SIZE as reported",
               "kernels": res}, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
