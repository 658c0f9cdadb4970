export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_lds -o lds -- python3 tools/chain_micro.py 5 > gpurun_out/pmc_lds/log.txt 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/pmc_lds/**/lds_counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void grk::(anonymous namespace)::", "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in per.items():
    if "conv_bf16" in k:
        print(f"{k:44s} dispatches {len(n[k]):3d}  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f}  (conflict cycles {c['SQ_LDS_BANK_CONFLICT'] / len(n[k]):.3e}, active {c['SQ_LDS_IDX_ACTIVE'] / len(n[k]):.3e} per dispatch)")
PY
