#!/bin/bash
# rocprofv3 kernel statistics of the training-side kernels (k_set_images, k_eval_batch, segmented sort, k_split_ord,
# k_negmine_windows, ...) plus counter passes for k_eval_batch. $1 = tag. Output under gpurun_out/train_<tag>/.
tag=${1:-t}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/train_$tag
mkdir -p $out
for job in "eval tools/bench_training_eval.py" "split tools/bench_split_search.py HAAR 20000" "negmine tools/bench_negmine.py 10 5"; do
  set -- $job; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 "$@" > $out/$name.log 2>&1
  tail -1 $out/$name.log
  python3 - $out/$name <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print("   %-60s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
for ctr in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" "WRITE_SIZE" "FETCH_SIZE"; do
  d=$out/pmc_$(echo $ctr | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE $ctr --output-format csv -d $d -- python3 tools/bench_training_eval.py > $d.log 2>&1
  python3 - $d <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "k_eval_batch" in k or "k_set_images" in k:
            print("  ", k[:50], " ".join("%s=%.2fM" % (c, sum(x) / len(x) / 1e6) for c, x in sorted(v.items())), "launches", len(next(iter(v.values()))))
PY
done
