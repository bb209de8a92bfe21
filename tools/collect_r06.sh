#!/bin/bash
# Round-6 evidence: the driver's bench command, the rocprofv3 kernel stats of the same command, MFMA-busy counters of the epoch's
# kernels, the optimizer launch's workgroup stamps, the vectoriser's HBM / LDS counters at k = 6, 5 and 4.   bash tools/collect_r06.sh <out_dir>
out=${1:-gpurun_out/r06_f}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -z "$SKIP_BENCH" ]; then
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time
echo "bench rc=$? $(grep real $out/bench_default.time)"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fixed-job --no-k-sweep --no-fp32-form --no-cfg5 > $out/trace_bench.json 2> $out/trace.err
cp $out/trace/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python3 - $out <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/bench_kernel_stats.csv")))
tot = 0; lines = ["== per-step kernels of `python3 bench.py --steps 5 --warmup 2` (rocprofv3 --kernel-trace --stats), us per launch"]
for r in rows:
    if int(r["Calls"]) >= 2000 and "flush_icache" not in r["Name"]:      # (TunableOp's helper while it selects the predict GEMMs: not a step kernel)
        tot += float(r["AverageNs"]); lines.append(f"   {r['Name'][:86]:86s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1000:7.2f}")
lines.append(f"   sum of the per-step kernels {tot/1000:.1f} us")
for r in rows:
    if any(k in r["Name"] for k in ("vectorise", "mimic", "col_stats", "stats", "mask_from", "rmsprop_kernel")) and int(r["Calls"]) < 2000:
        lines.append(f"   {r['Name'][:86]:86s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1000:7.2f}")
open(out + "/step_kernels.txt", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-fixed-job --no-k-sweep --no-fp32-form --no-cfg5 > $out/pmc_mfma.json 2> $out/pmc_mfma.err
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc_mfma/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
def key(n):
    for k, v in (("Cijk_", "hipBLASLt/rocBLAS GEMMs (Cijk_*)"), ("wgrad_rmsprop_kernel", "wgrad_rmsprop_kernel (dW1 tiles + optimizer)"), ("wgrad_q16", "wgrad_q16_kernel (dW1 tiles + RMSprop on W1)"),
                 ("l1_rms", "l1_rms_kernel (own layer-1 tiles + the previous step's optimizer tail)"), ("l1_fwd", "l1_fwd_kernel (own layer-1 tiles)"),
                 ("l1_planes", "l1_planes_kernel (layer 1 on the fp16 matrix cores from two-plane operands)"), ("wgrad_dplanes", "wgrad_dplanes_rms_kernel (dW1 from dr1's and the batch's planes by LDS-DMA + RMSprop on W1 + W1's planes + the step's optimizer tail on the loader waves)"),
                 ("reduce_rms", "reduce_rms_kernel (sum of the K-slice partials)"), ("mid_fwd", "mid_fwd_kernel"),
                 ("mid_bwd", "mid_bwd_kernel"), ("nce_pass1", "nce_pass1_kernel"), ("nce_pass2", "nce_pass2_kernel"), ("vectorise3", "vectorise3_kernel")):
        if k in n: return v
    return None
for r in csv.DictReader(open(f)):
    k = key(r["Kernel_Name"])
    if k is None: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
res = {}
for k, c in acc.items():
    n = len(seen[k]); d = {x: v / n for x, v in c.items()}; d["dispatches"] = n
    if d.get("GRBM_GUI_ACTIVE"): d["mfma_busy_fraction_of_kernel"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)
    res[k] = d
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 (cfg2); means per dispatch; "
                     "mfma_busy_fraction = busy cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)", "kernels": res}, open(out + "/epoch_pmc_mfma.json", "w"), indent=1)
print(json.dumps({k: round(v.get("mfma_busy_fraction_of_kernel", 0), 3) for k, v in res.items()}))
PY
bash tools/trace_step_c200.sh $out/c200 > $out/step_kernels_c200.txt 2>&1; tail -25 $out/step_kernels_c200.txt
find $out -name "*kernel_trace.csv" -delete
for K in 6 5 4; do
  K=$K bash tools/pmc_vectorise.sh $out/pmc_vec_k$K $out/vectorise_pmc_k$K.json > $out/vectorise_pmc_summary_k$K.txt 2>&1
  tail -3 $out/vectorise_pmc_summary_k$K.txt
done
# gpurun copies back at most 64 MiB: the raw traces / counter dumps are summarised above, only the summaries travel
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
du -sh $out
