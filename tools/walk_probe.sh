#!/bin/bash
# Experiment (round 3): does an XCD keep a strip's x panels in its L2 when it walks along N (MUMPY_WS_WALK=1, tuning build), and do
# non-temporal output stores (libmumpy_hip_nt.so: `make -C <pkg>/csrc nt`, -DMUMPY_WS_STORE_AUX=2) help it?  FETCH_SIZE per launch + duration.
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd/lib
python3 - <<'PY' || exit 1
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = '''
import sys, os, torch
sys.path[:0] = [os.environ["R"], os.environ["R"] + "/multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd"]
from mumpy_hip import ops
for m, n, k in ((8192, 2048, 512), (16484, 1536, 512), (8192, 512, 2048), (7840, 2048, 512)):
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
    y = ops.linear(x, w, b); ref = (x.double() @ w.double().t() + b.double())
    print(m, n, k, float((y.double() - ref).abs().max() / ref.abs().max()))
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-5
'''
P = R + "/multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd/lib"
for lib in ("libmumpy_hip_tuning.so", "libmumpy_hip_nt.so"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, R=R, MUMPY_HIP_LIB=P + "/" + lib, MUMPY_WS_WALK="1"), check=True)
print("walk / nt variants correct")
PY
for shape in "8192 2048 512 1" "8192 1536 512 0" "8192 512 2048 0"; do
  set -- $shape
  for lib in tuning nt; do for walk in 0 1; do
    tag=walk_$1_$2_$3_${lib}_w$walk
    MUMPY_HIP_LIB=$P/libmumpy_hip_$lib.so MUMPY_WS_WALK=$walk bash tools/pmc.sh $tag FETCH_SIZE -- linear $1 $2 $3 $4 > /dev/null || exit 1
    MUMPY_HIP_LIB=$P/libmumpy_hip_$lib.so MUMPY_WS_WALK=$walk python3 tools/kernel_micro.py linear $1 $2 $3 $4 > gpurun_out/$tag.time
    python3 - $tag <<'PY'
import csv, glob, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); tag = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/{tag}/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_ws" in r["Kernel_Name"]]
print(tag, f"read {sum(v) / len(v) * 2 * 1024 / 1e6:.1f} MB/launch;", open(f"{R}/gpurun_out/{tag}.time").read().strip().splitlines()[-1])
PY
  done; done
done
