#!/bin/bash
# GPU box: build the sampler and record the shader clock during x3 / bf16 paper-size steps (tools/clock_trace.py) -> gpurun_out/clock_*.json
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/probes/clock_probe.hip -o /tmp/libclockprobe.so
mkdir -p gpurun_out
for p in ${PRECISIONS:-x3 bf16}; do
  python tools/clock_trace.py --config ${CONFIG:-paper} --precision $p --out gpurun_out/clock_${CONFIG:-paper}_$p.json 2>&1 | grep -v amdgpu.ids | tail -1
done
