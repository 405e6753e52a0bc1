#!/bin/bash
# Round-4 record of the multi-rank flow on ONE GPU (profiles/r4_multirank_*): bench.py with N ranks sharing device 0 (gloo: the
# collective staged through the host) and the one-rank RCCL run.  At most 6 processes may use the card of a test box, so the
# largest world is 5 (the ranks plus the elastic agent that starts them: a 6-rank run was ended by the guard with 7 processes on the card).  Usage (on the GPU box): bash tests/tools_multirank_log.sh
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4_multirank
mkdir -p $OUT
export VOLREN_SAMPLE_POOL_MB=2048
run() {   # name, then the bench arguments
    local name=$1; shift
    echo "== $name: python bench.py $* (VOLREN_DIST_BACKEND=${VOLREN_DIST_BACKEND:-nccl})" | tee $OUT/$name.log
    python bench.py "$@" --cpu-budget 0 --extra-configs none >> $OUT/$name.log 2>&1
    grep '^{' $OUT/$name.log | tail -1 > $OUT/$name.json
    python - "$OUT/$name.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("   n_gpus %d  rccl_ranks %d  backend %s  crc %08x  value %.1f  pipelined %s Msamples/s" % (
    j["n_gpus"], j["rccl_ranks"], j["dist_backend"], j["frame_crc32"], j["value"], ("%.1f" % j["value_pipelined"]) if j["value_pipelined"] else "-"))
PY
}
# small frame: N = 1, 2, 4 (what tests/test_gpu_multirank.py asserts)
run n1_256 --gpus 1 --width 256 --height 192 --spp 8 --steps 2 --warmup 1
VOLREN_DIST_BACKEND=gloo run n2_256_gloo --gpus 2 --width 256 --height 192 --spp 8 --steps 2 --warmup 1
VOLREN_DIST_BACKEND=gloo run n4_256_gloo --gpus 4 --width 256 --height 192 --spp 8 --steps 2 --warmup 1
# BASELINE configs[3]'s frame (1920x1080; 64 spp here), N = 1 and 5
run n1_1080 --gpus 1 --config c4:128 --width 1920 --height 1080 --spp 64 --steps 2 --warmup 1
VOLREN_DIST_BACKEND=gloo run n5_1080_gloo --gpus 5 --config c4:128 --width 1920 --height 1080 --spp 64 --steps 2 --warmup 1
# RCCL with one rank: init_process_group("nccl"), all_gather_into_tensor on the renderer's stream
run n1_256_rccl --gpus 1 --force-dist --width 256 --height 192 --spp 8 --steps 2 --warmup 1
run n1_1080_rccl --gpus 1 --force-dist --config c4:128 --width 1920 --height 1080 --spp 64 --steps 2 --warmup 1
python - <<'PY'
import glob, json
rows = {p.split("/")[-1][:-5]: json.load(open(p)) for p in sorted(glob.glob("gpurun_out/r4_multirank/*.json"))}
small = {k: v["frame_crc32"] for k, v in rows.items() if "256" in k}
big = {k: v["frame_crc32"] for k, v in rows.items() if "1080" in k}
print("256x192 CRCs:", {k: "%08x" % v for k, v in small.items()}, "ALL EQUAL" if len(set(small.values())) == 1 else "MISMATCH")
print("1920x1080 CRCs:", {k: "%08x" % v for k, v in big.items()}, "ALL EQUAL" if len(set(big.values())) == 1 else "MISMATCH")
assert len(set(small.values())) == 1 and len(set(big.values())) == 1
PY
