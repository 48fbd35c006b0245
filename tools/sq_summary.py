"""Mean SQ counter values per launch and kernel from the passes of tools/gpu_sq.sh (prints JSON)."""
import csv, glob, json, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "sq_pass*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print(json.dumps({k: {c: round(tot[k][c] / cnt[k][c], 1) for c in sorted(tot[k])} for k in sorted(tot)}, indent=1))
