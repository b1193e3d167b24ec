#!/bin/bash
# Run ON THE GPU BOX: dynamic VALU instruction count of the dominant kernel per wave and 4-pixel time step, for the main
# library and every prebuilt variant in gpurun_variants/ (one rocprofv3 --pmc pass each).  usage: valu_count.sh "<workloads>"   (default: the headline)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for wl in ${1:-cfg2_esim_f32_256x32x256x256_bilinear5}; do
  for lib in main $REPO/gpurun_variants/lib_*.so; do
    [ -e "$lib" ] || [ "$lib" = main ] || continue
    tag=$(basename $lib .so)
    rm -rf /tmp/vc_$tag
    if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$lib; fi
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/vc_$tag -o vc -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > /tmp/vc_$tag.json 2>/dev/null
    python3 - "$tag" "$wl" /tmp/vc_$tag /tmp/vc_$tag.json << 'PY'
import csv, glob, json, sys
tag, wl, d, jf = sys.argv[1:5]
acc = {}
n = 0
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "voxel_kernel" in r["Kernel_Name"] and "shot_sum" not in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
try:
    line = json.loads(open(jf).read().strip().splitlines()[-1]); cfg = line["config"]; ms = line["roofline"]["kernel_ms_avg"]
    steps = cfg["frames"] - 1
except Exception as e:
    print(tag, wl, "bench failed", e); sys.exit(0)
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1)
print(f"{tag:10s} {wl[:22]:22s} {ms:7.4f} ms(prof)  VALU/wave-step {m.get('SQ_INSTS_VALU',0)/w/steps:6.1f}  SALU {m.get('SQ_INSTS_SALU',0)/w/steps:5.1f}  LDS {m.get('SQ_INSTS_LDS',0)/w/steps:4.1f}  wait_inst/wave_cycles {m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f}  busy_cycles {m.get('SQ_BUSY_CYCLES',0):.3g}")
PY
  done
done
