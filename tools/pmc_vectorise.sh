#!/bin/bash
# Counters of the vectorise kernel at cfg2 (100 000 x 10 kbp, k = 6, 4 views, device-drawn edits), one rocprofv3 pass per counter
# group (TCC: FETCH_SIZE takes 3 slots, WRITE_SIZE 2 -- separate passes; no trace domain together with --pmc on this pool).
#   bash tools/pmc_vectorise.sh <out_dir> [json]          (K=4|5|6 in the environment picks k; default 6)
# Writes <out_dir>/summary.txt and, when a second argument is given, the traffic summary bench.py reads (profiles/*_vectorise_pmc.json).
set -e
out=${1:-gpurun_out/pmc}
export IDELUCS_ABLATE_K=${K:-6}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
groups=("FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
        "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU")
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d $out/g$i -- python3 tools/ablate_vectorise.py 0 > $out/g$i.log 2>&1 || true
done
python3 - "$out" "$2" <<'PY'
import csv, glob, json, sys, collections
out, js = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        K = __import__("os").environ.get("IDELUCS_ABLATE_K", "6")
        if "vectorise3_kernel" in k: k = f"vectorise3_kernel<{K}>"
        elif "vectorise4_kernel" in k: k = f"vectorise4_kernel<{K}> (round 6: a wavefront per sequence, k = 4 / 5)"
        elif "vectorise2_kernel" in k: k = f"vectorise2_kernel<{K},false> (second pass: exits at once when v3 left nothing)"
        else: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in agg.items():
    lines.append(k)
    for c, v in sorted(cs.items()):
        lines.append(f"   {c:28s} launches={len(v):3d} mean per launch={sum(v)/len(v):.5g}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
K = int(__import__("os").environ.get("IDELUCS_ABLATE_K", "6"))
k3 = agg.get(f"vectorise3_kernel<{K}>", {}) or agg.get(f"vectorise4_kernel<{K}> (round 6: a wavefront per sequence, k = 4 / 5)", {})
if js and "FETCH_SIZE" in k3 and "WRITE_SIZE" in k3:
    f, w = sum(k3["FETCH_SIZE"]) / len(k3["FETCH_SIZE"]), sum(k3["WRITE_SIZE"]) / len(k3["WRITE_SIZE"])
    alg = 100000 * (2500 + 4 * 4 ** K * 4) / 1e9
    json.dump({"kernel": f"vectorise3_kernel<{K}>" if K >= 6 else f"vectorise4_kernel<{K}>", "workload": f"cfg2's input: 100000 x 10000 bp, k={K}, 4 views, device-drawn edits (tools/ablate_vectorise.py 0)",
               "algorithmic_gb_per_launch": round(alg, 4),
               "WRITE_SIZE_KB": w, "FETCH_SIZE_KB": f,
               "traffic_gb_per_launch": round((w + 2 * f) * 1024 / 1e9, 3),
               "note": "traffic = WRITE_SIZE + 2 x FETCH_SIZE: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads "
                       "(MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact for 16-byte-per-lane streaming stores.",
               "source": "tools/pmc_vectorise.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; mean over the launches of the run)"},
              open(js, "w"), indent=1)
PY
